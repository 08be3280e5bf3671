#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-s3l}; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_contraction.py tests/test_gpu_einsum.py -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
CUTENSOR_LOG_LEVEL=1 timeout 120 python bench.py --no-cpu > $O/bench_bal.log 2>&1
CUTENSOR_AMD_XCD_BALANCE=0 timeout 120 python bench.py --no-cpu > $O/bench_uni.log 2>&1
timeout 120 python bench.py --no-cpu > $O/bench_bal2.log 2>&1
CUTENSOR_AMD_XCD_BALANCE=0 timeout 120 python bench.py --no-cpu > $O/bench_uni2.log 2>&1
timeout 100 python tools/phase_timing.py --dump $O/t.npy | grep plan > $O/phase.jsonl
grep -i "xcd calibration" $O/bench_bal.log | head -2
