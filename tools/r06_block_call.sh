#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zzf; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_repack.py tests/test_gpu_h16.py -m gpu -x -q -k "permutation or repack or copied or in_place or sweep or fp32" > $O/pytest_block.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_block.log
python tools/bench_block_permute.py > $O/block_permute.jsonl 2>&1; cat $O/block_permute.jsonl
EINSUM_SHAPES_SET=sweep timeout 900 python tools/bench_einsum_shapes.py f32 > $O/sweep_shapes_f32.jsonl 2> $O/sweep_shapes_f32.err
EINSUM_SHAPES_SET=sweep timeout 900 python tools/bench_einsum_shapes.py bf16 > $O/sweep_shapes_bf16.jsonl 2> $O/sweep_shapes_bf16.err
export CTAMD_LIB_FLAVOUR=hooks
timeout 500 python tools/fuzz_elementwise.py > $O/fuzz_elementwise.log 2>&1; tail -1 $O/fuzz_elementwise.log | cut -c1-500
CUTENSOR_AMD_REPACK=f timeout 500 python tools/fuzz_contraction.py --cases 500 --seed 101 > $O/fuzz_default_copies_forced.log 2>&1; tail -1 $O/fuzz_default_copies_forced.log | cut -c1-500
CUTENSOR_AMD_REPACK=f timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 102 --sweep-k --strided > $O/fuzz_sweep_strided_copies_forced.log 2>&1; tail -1 $O/fuzz_sweep_strided_copies_forced.log | cut -c1-500
timeout 400 python tools/fuzz_einsum.py > $O/fuzz_einsum.log 2>&1; tail -1 $O/fuzz_einsum.log | cut -c1-300
