#!/bin/bash
# Round 5, sixth GPU call (PRODUCTION build): the whole GPU suite, smoke(), and the driver-style bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05f
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
echo "bench rc $?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05f/bench_steps20.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("value", d["value"], "frac", r.get("frac"), "frac_event_pair", r.get("frac_event_pair"), "mfma_busy", r.get("mfma_busy_pct"), "clock", r.get("sustained_clock_ghz"), "src", str(r.get("frac_source"))[:80])
    print("config", {k: d["config"].get(k) for k in ("cold_operands_gflops", "cold_operands_streamed_preference_gflops", "sample_protocol_gflops", "kernel_frac_live_trace", "mfma_busy_pct")})
    for s in d["secondary"]:
        print(" -", s.get("workload", "")[:90], "|", s.get("value"), s.get("unit"), "| frac", (s.get("roofline") or {}).get("frac"), "| busy", (s.get("roofline") or {}).get("mfma_busy_pct"), s.get("kernel"), s.get("error"))
except Exception as ex:
    print("parse failed", ex)
PY
tail -3 $OUT/bench_steps20.err
