#!/bin/bash
# SQ counters of the 16-bit kernels on the 8192^3 bench (separate --pmc run, no tracing).  usage: tools/h16_pmc.sh <tag> <layout> [env...]
TAG=$1; LAYOUT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
env "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc -o r -- python $ROOT/tools/bench_h16.py --layout $LAYOUT --reps 20 > $OUT/pmc.log 2>&1
cd $ROOT
f=$(find $OUT/pmc -name '*.db' | head -1)
[ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/pmc.summary.txt 2>&1
find $OUT -name '*.csv' -size +2M -delete
rm -rf $OUT/pmc
cat $OUT/pmc.summary.txt | head -40
