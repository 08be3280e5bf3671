#!/bin/bash
# Headline einsum with the three cache policies of the split-K partial stores (CUTENSOR_AMD_PARTIAL_STORE: write-through = default,
# plain = write-back, nt): bench line (whole step, GETT kernel alone, cold operands) twice each + a rocprofv3 kernel trace (the fold
# kernel's own time shows what the kernel boundary has to flush).  usage: tools/partial_store_policy.sh <outdir>
OUT=${1:-gpurun_out/partial_store}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
export TMPDIR=/tmp
cd $ROOT
for pol in wt plain nt; do
  for rep in 1 2; do
    CUTENSOR_AMD_PARTIAL_STORE=$pol python bench.py --no-secondary --no-pmc --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$pol', round(d['value']/1e3,2), 'TFLOP/s  step_us', round(d['ms_per_step']*1e3,2), ' gett_kernel_us', round(d['roofline']['mean_us'],2), ' cold_TFLOPs', round(d['cold_operands']['value']/1e3,1), ' sample_protocol_us', round(d['sample_protocol']['min_us'],1))"
  done
  (cd /tmp && CUTENSOR_AMD_PARTIAL_STORE=$pol rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace_$pol -o r -- python $ROOT/bench.py --steps 500 --warmup 50 --no-secondary --no-pmc --no-cpu > /dev/null 2>&1)
  f=$(find $ROOT/$OUT/trace_$pol -name '*.db' | head -1)
  [ -n "$f" ] && python tools/rocprof_summary.py $f 2>/dev/null | sed -n 2,4p
done
find $ROOT/$OUT -name '*.db' -delete; find $ROOT/$OUT -name '*.csv' -delete
