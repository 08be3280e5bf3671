#!/bin/bash
# Round 6, GPU call s: beta != 0 in the persistent kernel with three register sets of C chunks (requests further in front of the stores)
# against two sets (build/exp_c2/libcutensor.so, the build call r measured) and the one-tile kernel, alternating on one box; parity first.
set -u
OUT=gpurun_out/r06s; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_h16p.py -x -q > $OUT/h16p.log 2>&1; echo "h16p rc $?"; tail -5 $OUT/h16p.log
export CTAMD_LIB_FLAVOUR=hooks
V=$PWD/build/exp_c2/libcutensor.so
SH="8192,8192,8192;8192,8192,2048;8192,8192,1024"
for rep in 1 2; do
  for lay in mk,kn km,nk; do
    timeout 200 python tools/h16_shape_sweep.py --layout $lay --beta 0.5 --only "$SH" 2>/dev/null | sed 's/^{/{"lib": "depth3", /' >> $OUT/beta_depth_ab.jsonl
    CUTENSOR_AMD_LIBRARY=$V timeout 200 python tools/h16_shape_sweep.py --layout $lay --beta 0.5 --only "$SH" 2>/dev/null | sed 's/^{/{"lib": "depth2", /' >> $OUT/beta_depth_ab.jsonl
    CUTENSOR_AMD_H16P=0 timeout 200 python tools/h16_shape_sweep.py --layout $lay --beta 0.5 --only "$SH" 2>/dev/null | sed 's/^{/{"lib": "one_tile", /' >> $OUT/beta_depth_ab.jsonl
    timeout 200 python tools/h16_shape_sweep.py --layout $lay --beta 0.0 --only "$SH" 2>/dev/null | sed 's/^{/{"lib": "depth3", /' >> $OUT/beta_depth_ab.jsonl
  done
done
python - <<'PY'
import json, collections
r = collections.defaultdict(list)
for l in open("gpurun_out/r06s/beta_depth_ab.jsonl"):
    d = json.loads(l); r[(d["layout"], d["K"], d["beta"], d["lib"])].append(d["tflops"])
for k in sorted(r): print(k, r[k])
PY
