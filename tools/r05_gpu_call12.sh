#!/bin/bash
# Round 5, GPU call 12: the whole GPU suite on the production build (after the general-family tests were pinned to their family), the
# element-wise fuzzers with complex data.
set -u
OUT=gpurun_out/r05m; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/summary.txt; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -3
f() { name=$1; shift; timeout 400 python "$@" > $OUT/fuzz_$name.log 2>&1; echo "fuzz $name rc $? : $(tail -1 $OUT/fuzz_$name.log | cut -c1-260)" | tee -a $OUT/summary.txt; }
f elementwise tools/fuzz_elementwise.py --cases 500 --seed 17
f elementwise_wide tools/fuzz_elementwise.py --wide --cases 150 --seed 18
