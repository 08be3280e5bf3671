#!/bin/bash
# round 6, sweep-ragged K and repacked operands: fuzz sweeps (planner's choice; copies forced whenever the temporaries fit), then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zz7; mkdir -p $O
export TMPDIR=/tmp CTAMD_LIB_FLAVOUR=hooks
timeout 500 python tools/fuzz_contraction.py --cases 500 --seed 81 --sweep-k > $O/fuzz_sweep.log 2>&1; tail -1 $O/fuzz_sweep.log | cut -c1-600
timeout 500 python tools/fuzz_contraction.py --cases 300 --seed 82 --sweep-k --strided > $O/fuzz_sweep_strided.log 2>&1; tail -1 $O/fuzz_sweep_strided.log | cut -c1-600
CUTENSOR_AMD_REPACK=f timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 83 --sweep-k > $O/fuzz_sweep_copies_forced.log 2>&1; tail -1 $O/fuzz_sweep_copies_forced.log | cut -c1-600
CUTENSOR_AMD_REPACK=f timeout 500 python tools/fuzz_contraction.py --cases 300 --seed 84 --strided > $O/fuzz_default_copies_forced.log 2>&1; tail -1 $O/fuzz_default_copies_forced.log | cut -c1-600
CUTENSOR_AMD_REPACK=f timeout 400 python tools/fuzz_einsum.py > $O/fuzz_einsum_copies_forced.log 2>&1; tail -1 $O/fuzz_einsum_copies_forced.log | cut -c1-300
unset CTAMD_LIB_FLAVOUR
[ -n "$SKIP_PYTEST" ] || timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
