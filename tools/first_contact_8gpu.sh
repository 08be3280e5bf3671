#!/bin/bash
# First contact with a multi-GPU node (none is reachable from the build container: every N > 1 path of this repository has only
# run on plan-only handles, on logical devices of one GPU, and through the CPU replay of tests/test_mg_replay_cpu.py).
# One command; everything lands under profiles/<tag>_* (tracked) and gpurun_out/<tag>/ (scratch):
#   1. the GPU tests of the multi-device libraries on the devices that are visible
#   2. bench.py --gpus N for N = 1 2 4 8 (the driver's own scaling sequence), default transport (auto: timed all-gather vs send/recv)
#   3. bench.py --gpus 8 with each cuTENSORMg transport pinned: allgather, sendrecv, peer (hipMemcpyPeerAsync, no RCCL)
#   4. bench.py --gpus 8 with the gather cut into 1 / 2 / 4 waves (CUTENSORMG_AMD_WAVES)
#   5. rocprofv3 kernel trace of one --gpus 8 run (per-kernel times per device; no counters: --pmc never rides with tracing)
#   6. the reference's own samples, unmodified: contraction_multi_gpu (8 devices), blog_post 8 <scaling 1..4>
# usage: tools/first_contact_8gpu.sh [--dry-run] [tag] [steps]
#   --dry-run: every command is printed ("+ ...") instead of executed, for FIRST_CONTACT_NGPU (default 8) visible GPUs, nothing is
#   written outside a temporary directory — tests/test_tools_compile_cpu.py walks the script this way, so that the script itself is
#   not first-contact code (every file it names must exist, every bench.py flag must parse).
set -u
DRY=0
if [ "${1:-}" = "--dry-run" ]; then DRY=1; shift; fi
TAG=${1:-first8}
STEPS=${2:-50}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
PROF=$ROOT/profiles
if [ "$DRY" = 1 ]; then OUT=$(mktemp -d); PROF=$OUT/profiles; fi
mkdir -p "$OUT" "$PROF"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd "$ROOT"
if [ "$DRY" = 1 ]; then NGPU=${FIRST_CONTACT_NGPU:-8}; else NGPU=$(python -c 'import torch; print(torch.cuda.device_count())'); fi
echo "visible GPUs: $NGPU" | tee "$OUT/summary.txt"
x() {   # x <command...>: run it, or (dry run) print it; redirections stay with the caller and receive nothing in a dry run
  if [ "$DRY" = 1 ]; then echo "+ $*" >&3; else "$@"; fi
}
exec 3>&1

run_bench() {   # run_bench <name> <gpus> [env...]
  local name=$1 n=$2; shift 2
  if [ "$n" -gt 1 ]; then
    x env "$@" timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) \
        bench.py --gpus "$n" --steps "$STEPS" --warmup 5 > "$OUT/$name.json" 2> "$OUT/$name.err"
  else
    x env "$@" timeout 1200 python bench.py --gpus 1 --steps "$STEPS" --warmup 5 > "$OUT/$name.json" 2> "$OUT/$name.err"
  fi
  [ "$DRY" = 1 ] && return 0
  python - "$OUT/$name.json" "$name" >> "$OUT/summary.txt" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d.get("config", {})
    print("%-28s n_gpus %s value %.0f %s  mg_value %s  speedup_vs_1 %s  transport/workload: %s" % (
        sys.argv[2], d.get("n_gpus"), d.get("value", 0), d.get("unit"), d.get("mg_value"), d.get("speedup_vs_1_same_workload"), str(c.get("workload"))[:110]))
except Exception as ex:
    print("%-28s FAILED to parse: %s" % (sys.argv[2], ex))
PY
  cp "$OUT/$name.json" "$PROF/${TAG}_$name.json" 2>/dev/null
}

# 1. tests
x timeout 1800 python -m pytest tests/test_gpu_mg.py tests/test_gpu_mp.py tests/test_gpu_samples.py -q > "$OUT/gpu_tests_multi.log" 2>&1
tail -3 "$OUT/gpu_tests_multi.log" | tee -a "$OUT/summary.txt"
# 2. the scaling sequence
for n in 1 2 4 8; do [ "$n" -le "$NGPU" ] && run_bench "bench_gpus$n" "$n"; done
if [ "$NGPU" -ge 2 ]; then
  N=$NGPU; [ "$N" -gt 8 ] && N=8
  # 3. transports
  for t in allgather sendrecv peer; do run_bench "bench_gpus${N}_transport_$t" "$N" CUTENSORMG_AMD_TRANSPORT=$t; done
  # 4. gather waves
  for w in 1 2 4; do run_bench "bench_gpus${N}_waves_$w" "$N" CUTENSORMG_AMD_WAVES=$w; done
  # 5. kernel trace
  (cd /tmp && x timeout 1200 rocprofv3 --kernel-trace --stats -d "$OUT/trace_gpus$N" -o r -- python "$ROOT/bench.py" --gpus "$N" --steps 10 --warmup 2 > "$OUT/trace_gpus$N.log" 2>&1)
  f=$(find "$OUT/trace_gpus$N" -name '*.db' 2>/dev/null | head -1)
  [ "$DRY" = 1 ] && f=trace.db
  [ -n "$f" ] && x python tools/rocprof_summary.py "$f" > "$PROF/${TAG}_trace_gpus$N.summary.txt" 2>&1
  # 6. the reference's samples, unmodified (built by oracle/build_ref_samples.sh into oracle/_ref/)
  for exe in contraction_multi_gpu; do [ -x oracle/_ref/$exe ] && (x timeout 600 oracle/_ref/$exe > "$OUT/ref_$exe.log" 2>&1; tail -2 "$OUT/ref_$exe.log" | tee -a "$OUT/summary.txt"); done
  for s in 1 2 3 4; do [ -x oracle/_ref/blog_post ] && (x timeout 600 oracle/_ref/blog_post "$N" "$s" > "$OUT/ref_blog_post_${N}_$s.log" 2>&1; tail -1 "$OUT/ref_blog_post_${N}_$s.log" | tee -a "$OUT/summary.txt"); done
fi
cp "$OUT/summary.txt" "$PROF/${TAG}_summary.txt"
find "$OUT" -name '*.db' -delete 2>/dev/null
cat "$OUT/summary.txt"
[ "$DRY" = 1 ] && rm -rf "$OUT"
exit 0
