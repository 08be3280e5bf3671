#!/bin/bash
O=gpurun_out/${1:-s3c}; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_contraction.py tests/test_gpu_einsum.py -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
CUTENSOR_AMD_ABLATION=1 timeout 300 python tools/tune_gett.py --problem einsum --splits 256 --kernels 54,68,69,70,71 --max 8 --reps 200 2>&1 | grep rank > $O/tune.jsonl
for a in 54 70 71; do CUTENSOR_AMD_ABLATION=1 CUTENSOR_AMD_FORCE=$a:256 python tools/phase_timing.py 2>&1 | grep plan; done > $O/phase.jsonl
