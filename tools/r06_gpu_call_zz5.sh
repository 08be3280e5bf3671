#!/bin/bash
# Round 6, after the fp32 row epilogue: fuzz sweeps (fp32-heavy: default, aligned, strided, ring-kernel candidates, einsum front end,
# cuTENSORMg / cutensorMp), driver-style bench lines.
set -u
OUT=gpurun_out/r06zz5; mkdir -p $OUT
export TMPDIR=/tmp CTAMD_LIB_FLAVOUR=hooks
timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 71 > $OUT/fuzz_default.log 2>&1; tail -1 $OUT/fuzz_default.log | cut -c1-200
timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 72 --aligned > $OUT/fuzz_aligned.log 2>&1; tail -1 $OUT/fuzz_aligned.log | cut -c1-200
timeout 500 python tools/fuzz_contraction.py --cases 300 --seed 73 --strided > $OUT/fuzz_strided.log 2>&1; tail -1 $OUT/fuzz_strided.log | cut -c1-200
timeout 500 python tools/fuzz_contraction.py --cases 200 --seed 74 --strided --all-types > $OUT/fuzz_strided_all.log 2>&1; tail -1 $OUT/fuzz_strided_all.log | cut -c1-200
timeout 500 python tools/fuzz_stream.py --cases 160 --seed 75 --ranks 8 > $OUT/fuzz_stream.log 2>&1; tail -1 $OUT/fuzz_stream.log | cut -c1-200
timeout 400 python tools/fuzz_einsum.py > $OUT/fuzz_einsum.log 2>&1; tail -1 $OUT/fuzz_einsum.log | cut -c1-200
timeout 400 python tools/fuzz_mg.py > $OUT/fuzz_mg.log 2>&1; tail -1 $OUT/fuzz_mg.log | cut -c1-200
timeout 400 python tools/fuzz_mp.py > $OUT/fuzz_mp.log 2>&1; tail -1 $OUT/fuzz_mp.log | cut -c1-200
unset CTAMD_LIB_FLAVOUR
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2>&1; tail -1 $OUT/bench_steps20.log | cut -c1-300
python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log | cut -c1-300
