#!/bin/bash
# Ablation of the default (ping-pong) 16-bit kernel on U(-1,1) data: what each kind of data movement costs under the power
# limit (measurement only; wrong results).  usage: tools/h16_pp_abl.sh <outfile>
OUT=${1:-gpurun_out/h16_pp_abl.jsonl}
for z in "" "--zeros"; do
for a in 0 1 2 4 3 5; do
    CUTENSOR_AMD_H16_ABL=$a timeout 120 python tools/bench_h16.py --layout km,kn $z 2>&1 | grep workload | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'abl': $a, 'zeros': '$z' != '', 'tflops': d['tflops'], 'ms': d['ms_per_call']}))" >> $OUT
done; done
python - <<PY
import ctypes, json, sys
sys.path.insert(0, '.')
from cudalibrarysamples_amd import cutensor as ct
import torch
torch.cuda.init()
out = {}
for name, kind in (('zeros', 0), ('uniform', 1)):
    v = ctypes.c_float(0); ct.lib.ctamdMeasureMfmaCeiling(1, kind, ctypes.byref(v)); out[name] = v.value
print(json.dumps({'mfma_only': out}))
PY
