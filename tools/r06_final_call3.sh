#!/bin/bash
# Round 6, last call: the whole GPU suite, every fuzz sweep, the driver-style bench lines on the final tree.
set -u
TAG=${1:-r06zzl}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
export CTAMD_LIB_FLAVOUR=hooks
timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 171 > $OUT/fuzz_default.log 2>&1; tail -1 $OUT/fuzz_default.log | cut -c1-200
timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 172 --aligned > $OUT/fuzz_aligned.log 2>&1; tail -1 $OUT/fuzz_aligned.log | cut -c1-200
timeout 500 python tools/fuzz_contraction.py --cases 300 --seed 173 --strided > $OUT/fuzz_strided.log 2>&1; tail -1 $OUT/fuzz_strided.log | cut -c1-200
timeout 500 python tools/fuzz_contraction.py --cases 200 --seed 174 --strided --all-types > $OUT/fuzz_strided_all.log 2>&1; tail -1 $OUT/fuzz_strided_all.log | cut -c1-200
timeout 500 python tools/fuzz_contraction.py --cases 300 --seed 175 --many-modes > $OUT/fuzz_many_modes.log 2>&1; tail -1 $OUT/fuzz_many_modes.log | cut -c1-200
timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 176 --sweep-k --strided > $OUT/fuzz_sweep_strided.log 2>&1; tail -1 $OUT/fuzz_sweep_strided.log | cut -c1-200
CUTENSOR_AMD_REPACK=f timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 177 --all-types > $OUT/fuzz_all_types_copies_forced.log 2>&1; tail -1 $OUT/fuzz_all_types_copies_forced.log | cut -c1-200
timeout 500 python tools/fuzz_stream.py --cases 160 --seed 178 --ranks 8 > $OUT/fuzz_stream.log 2>&1; tail -1 $OUT/fuzz_stream.log | cut -c1-200
timeout 400 python tools/fuzz_einsum.py > $OUT/fuzz_einsum.log 2>&1; tail -1 $OUT/fuzz_einsum.log | cut -c1-200
timeout 400 python tools/fuzz_elementwise.py > $OUT/fuzz_elementwise.log 2>&1; tail -1 $OUT/fuzz_elementwise.log | cut -c1-200
timeout 400 python tools/fuzz_mg.py > $OUT/fuzz_mg.log 2>&1; tail -1 $OUT/fuzz_mg.log | cut -c1-200
timeout 400 python tools/fuzz_mp.py > $OUT/fuzz_mp.log 2>&1; tail -1 $OUT/fuzz_mp.log | cut -c1-200
unset CTAMD_LIB_FLAVOUR
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2>&1; tail -1 $OUT/bench_steps20.log | cut -c1-300
python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log | cut -c1-300
