#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-s3j}; mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_h16.py -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for l in mk,kn km,kn mk,nk; do timeout 120 python tools/bench_h16.py --layout $l 2>&1 | grep workload; done > $O/bench_h16.jsonl
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/pmc_sq -o r -- python $GRAFT_REPO_ROOT/tools/bench_h16.py --reps 5 > $O/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/pmc_sq -name '*.db' | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $O/pmc_sq.summary.txt 2>&1
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -size +8M -delete
