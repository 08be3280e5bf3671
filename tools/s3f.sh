#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-s3f}; mkdir -p $O
cd $GRAFT_REPO_ROOT
for a in 54; do CUTENSOR_AMD_ABLATION=1 CUTENSOR_AMD_FORCE=$a:256 python tools/phase_timing.py 2>&1 | grep plan; done > $O/phase.jsonl
python bench.py --no-cpu > $O/bench.log 2>&1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/trace -name '*.db' | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $O/trace.summary.txt 2>&1
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -size +8M -delete
