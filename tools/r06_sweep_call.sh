#!/bin/bash
# round 6, sweep-ragged K: parity of the new path, then its rate beside the general family and the vendor BLAS
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zzk; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_h16.py -m gpu -x -q -k "sweep or ragged or unaligned or copied or in_place" > $O/pytest_sweep.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_sweep.log
tail -5 $O/pytest_sweep.log
EINSUM_SHAPES_SET=sweep timeout 600 python tools/bench_einsum_shapes.py bf16 > $O/sweep_shapes_bf16.jsonl 2> $O/sweep_shapes_bf16.err
CTAMD_LIB_FLAVOUR=hooks EINSUM_SHAPES_SET=sweep CUTENSOR_AMD_GEN=f timeout 600 python tools/bench_einsum_shapes.py bf16 > $O/sweep_shapes_bf16_general_family.jsonl 2> $O/sweep_shapes_gen.err
cat $O/sweep_shapes_bf16.jsonl $O/sweep_shapes_bf16_general_family.jsonl | cut -c1-400
python tools/bench_unaligned.py --shapes "4096,4096,4096;4096,4096,4104;4100,4100,4100;4097,4097,4097;2048,2048,200" > $O/unaligned_bf16_after_sweep_mask.jsonl 2>&1
cut -c1-230 $O/unaligned_bf16_after_sweep_mask.jsonl
CTAMD_LIB_FLAVOUR=hooks timeout 500 python tools/fuzz_contraction.py --cases 600 --seed 91 --sweep-k > $O/fuzz_sweep.log 2>&1; tail -1 $O/fuzz_sweep.log | cut -c1-600
CTAMD_LIB_FLAVOUR=hooks timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 92 --sweep-k --strided > $O/fuzz_sweep_strided.log 2>&1; tail -1 $O/fuzz_sweep_strided.log | cut -c1-600
CTAMD_LIB_FLAVOUR=hooks timeout 500 python tools/fuzz_contraction.py --cases 400 --seed 93 --ragged-k > $O/fuzz_ragged.log 2>&1; tail -1 $O/fuzz_ragged.log | cut -c1-600
