#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-s3i}; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_contraction.py tests/test_gpu_einsum.py tests/test_gpu_h16.py -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 120 python bench.py --no-cpu > $O/bench_fused.log 2>&1
CUTENSOR_AMD_FUSED_FOLD=0 timeout 120 python bench.py --no-cpu > $O/bench_unfused.log 2>&1
timeout 100 python tools/phase_timing.py --dump $O/t.npy | grep plan > $O/phase.jsonl
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name '*.db' | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $O/trace.summary.txt 2>&1
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -size +8M -delete
