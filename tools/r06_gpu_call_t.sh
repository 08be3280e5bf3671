#!/bin/bash
# Round 6, GPU call t: the whole GPU suite on the current tree, the unaligned / ragged / strided fuzz sweeps, driver-style bench line.
set -u
OUT=gpurun_out/r06t; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -25 $OUT/pytest_gpu.log
export CTAMD_LIB_FLAVOUR=hooks
timeout 400 python tools/fuzz_contraction.py --cases 300 --seed 61 > $OUT/fuzz_default.log 2>&1; tail -2 $OUT/fuzz_default.log
timeout 400 python tools/fuzz_contraction.py --cases 200 --seed 62 --strided --all-types > $OUT/fuzz_strided_all.log 2>&1; tail -2 $OUT/fuzz_strided_all.log
timeout 400 python tools/fuzz_contraction.py --cases 200 --seed 63 --ragged-k > $OUT/fuzz_ragged_k.log 2>&1; tail -2 $OUT/fuzz_ragged_k.log
unset CTAMD_LIB_FLAVOUR
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/time.txt; tail -c 600 $OUT/bench.json; cat $OUT/time.txt
