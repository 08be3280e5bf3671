#!/bin/bash
# Round 6, GPU call v: bits of the persistent kernel's beta path against the one-tile twin (fp16: one v_fma_mixlo_f16 per element in both);
# bandwidth of the tiled wide-element kernels (fp64 / complex64 / complex128 permutations and reductions, 16-bit reductions).
set -u
OUT=gpurun_out/r06v; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_h16p.py -x -q > $OUT/h16p.log 2>&1; echo "h16p rc $?"; tail -4 $OUT/h16p.log
: > $OUT/bandwidth_wide.jsonl
timeout 300 python tools/bench_bandwidth.py --dtype c64 --n 1024 >> $OUT/bandwidth_wide.jsonl 2>$OUT/err.log
timeout 300 python tools/bench_bandwidth.py --dtype f64 --ext 2048,2048,1024 >> $OUT/bandwidth_wide.jsonl 2>>$OUT/err.log
timeout 300 python tools/bench_bandwidth.py --dtype c128 --n 1024 >> $OUT/bandwidth_wide.jsonl 2>>$OUT/err.log
timeout 300 python tools/bench_bandwidth.py --dtype bf16 --n 2048 >> $OUT/bandwidth_wide.jsonl 2>>$OUT/err.log
cut -c1-330 $OUT/bandwidth_wide.jsonl
