#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-s3m}; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_contraction.py tests/test_gpu_einsum.py -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 120 python bench.py --no-cpu > $O/bench1.log 2>&1
timeout 120 python bench.py --no-cpu > $O/bench2.log 2>&1
for i in 1 2; do timeout 100 python tools/phase_timing.py | grep plan; done > $O/phase.jsonl
