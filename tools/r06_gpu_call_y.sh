#!/bin/bash
# Round 6, GPU call y: every beta path of the 16-bit LDS-DMA family on one single-rounding helper (h_round16_with_c): parity incl. the bits
# of the two 256 x 256 twins; general family: split-K from 16 K-tiles again (the reference's 'mlik,lkjm->lij').
set -u
OUT=gpurun_out/r06y; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_h16p.py tests/test_gpu_h16.py tests/test_gpu_h16_unaligned.py tests/test_gpu_gen.py tests/test_gpu_einsum.py tests/test_gpu_ref_torch_binding.py -x -q > $OUT/h16.log 2>&1
grep -n "AssertionError" $OUT/h16.log | cut -c1-900; tail -2 $OUT/h16.log
timeout 600 python tools/bench_gen.py h16 einsum > $OUT/bench_gen.jsonl 2>$OUT/err.log; cut -c1-300 $OUT/bench_gen.jsonl
