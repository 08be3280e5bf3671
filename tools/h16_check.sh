#!/bin/bash
# GPU check of the 16-bit GETT kernel: parity tests (three passes: races show up as rare wrong tiles), then the
# 8192^3 bench in three operand layouts.  usage: tools/h16_check.sh <tag>
O=gpurun_out/${1:-h16}; mkdir -p $O
for i in 1 2 3; do ( timeout 600 python -m pytest tests/test_gpu_h16.py -x -q ) 2>&1 | tail -2; done | tee $O/pytest.log
for l in mk,kn km,kn mk,nk; do timeout 120 python tools/bench_h16.py --layout $l 2>&1 | grep workload; done > $O/bench_h16.jsonl
python - <<PY
import json
for l in open("$O/bench_h16.jsonl"):
    d=json.loads(l); print(d["workload"], "loop ms %.3f (%.0f TF) kern %.3f min %.3f TF %.0f frac %.3f"%(d["ms_per_call"],d["tflops"],d["kernel_mean_ms"],d["kernel_min_ms"],d["kernel_tflops"],d["frac_of_bf16_mfma_peak"]))
PY
