#!/bin/bash
# Collect the round's evidence on the GPU box: rocprofv3 kernel-trace stats of bench.py, then the PMC
# passes (separate runs, --pmc only) for the dominant kernel.  Output under gpurun_out/<tag>/.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-secondary --no-cold $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o r -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o r -- $BENCH > $OUT/pmc_write.log 2>&1
cd $ROOT
for d in trace pmc_sq pmc_fetch pmc_write; do
  f=$(find $OUT/$d -name '*.db' | head -1)
  [ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/$d.summary.txt 2>&1
done
python tools/rocprof_summary.py --traffic-json $OUT/pmc_traffic_einsum.json $(find $OUT/pmc_fetch -name '*.db' | head -1) $(find $OUT/pmc_write -name '*.db' | head -1) > /dev/null 2>&1
find $OUT -name '*.csv' -size +2M -delete
ls -la $OUT
