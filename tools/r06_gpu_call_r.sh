#!/bin/bash
# Round 6, GPU call r: beta != 0 in the persistent 16-bit kernel's streaming epilogue (C through the idle row image): parity (bits of the
# one-tile twin), A/B against the one-tile kernel (CUTENSOR_AMD_H16P=0 = what a beta != 0 call ran before) on one box; cutensorMg with
# the cross-device join issued by the workers: parity + host cost.
set -u
OUT=gpurun_out/r06r; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_h16p.py -x -q > $OUT/h16p.log 2>&1; echo "h16p rc $?"; tail -5 $OUT/h16p.log
timeout 900 python -m pytest tests/test_gpu_mg.py -x -q > $OUT/mg.log 2>&1; echo "mg rc $?"; tail -3 $OUT/mg.log
export CTAMD_LIB_FLAVOUR=hooks
for rep in 1 2; do
  for lay in mk,kn km,kn mk,nk; do
    for beta in 0.5 0.0; do
      timeout 200 python tools/h16_shape_sweep.py --layout $lay --beta $beta --only "8192,8192,8192;8192,8192,2048;8192,8192,1024" 2>/dev/null | sed 's/^{/{"lib": "persistent", /' >> $OUT/beta_ab.jsonl
      CUTENSOR_AMD_H16P=0 timeout 200 python tools/h16_shape_sweep.py --layout $lay --beta $beta --only "8192,8192,8192;8192,8192,2048;8192,8192,1024" 2>/dev/null | sed 's/^{/{"lib": "one_tile", /' >> $OUT/beta_ab.jsonl
    done
  done
done
python - <<'PY'
import json, collections
r = collections.defaultdict(list)
for l in open("gpurun_out/r06r/beta_ab.jsonl"):
    d = json.loads(l); r[(d["layout"], d["K"], d["beta"], d["lib"])].append(d["tflops"])
for k in sorted(r): print(k, r[k])
PY
unset CTAMD_LIB_FLAVOUR
timeout 300 python tools/mg_host_cost_n.py 512 > $OUT/host_cost_threads.jsonl 2>$OUT/err1.log; cut -c1-400 $OUT/host_cost_threads.jsonl
CTAMD_LIB_FLAVOUR=hooks CUTENSORMG_AMD_THREADS=0 timeout 300 python tools/mg_host_cost_n.py 512 > $OUT/host_cost_single_thread.jsonl 2>$OUT/err2.log; cut -c1-400 $OUT/host_cost_single_thread.jsonl
nproc
