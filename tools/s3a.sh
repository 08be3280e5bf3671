#!/bin/bash
# session-3 first GPU call: ground truth of the restored tree
O=gpurun_out/s3a; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
CUTENSOR_AMD_ABLATION=1 timeout 300 python tools/tune_gett.py --problem einsum --max 30 --reps 100 > $O/tune_einsum.jsonl 2>&1
timeout 300 python bench.py > $O/bench.log 2>&1
tail -2 $O/bench.log
timeout 100 python tools/phase_timing.py > $O/phase.log 2>&1
CUTENSOR_AMD_ABLATION=1 timeout 200 python tools/tune_gett.py --problem contraction --max 12 --reps 20 > $O/tune_contraction.jsonl 2>&1
