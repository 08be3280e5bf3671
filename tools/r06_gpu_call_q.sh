#!/bin/bash
# Round 6, GPU call q: cutensorMgContraction with the batched cell copy (mg_kernels.hip) and per-device workers: parity, host cost at
# 1 / 2 / 4 / 8 logical devices with and without the workers, mg fuzz; the reference's fp16 case 'mlik,lkjm->lij' (bench_gen.py).
set -u
OUT=gpurun_out/r06q; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mg.py tests/test_gpu_samples.py -x -q > $OUT/mg.log 2>&1; echo "mg rc $?"; tail -3 $OUT/mg.log
timeout 300 python tools/mg_host_cost_n.py 512 > $OUT/host_cost_threads.jsonl 2>$OUT/err1.log; cat $OUT/host_cost_threads.jsonl | cut -c1-330
CTAMD_LIB_FLAVOUR=hooks CUTENSORMG_AMD_THREADS=0 timeout 300 python tools/mg_host_cost_n.py 512 > $OUT/host_cost_single_thread.jsonl 2>$OUT/err2.log; cut -c1-330 $OUT/host_cost_single_thread.jsonl
timeout 300 python tools/mg_host_cost_n.py 4096 > $OUT/host_cost_threads_4096.jsonl 2>>$OUT/err1.log; cut -c1-330 $OUT/host_cost_threads_4096.jsonl
timeout 300 python tools/fuzz_mg.py > $OUT/fuzz_mg.log 2>&1; tail -2 $OUT/fuzz_mg.log
timeout 300 python tools/bench_gen.py > $OUT/bench_gen.jsonl 2>$OUT/err3.log; grep mlik $OUT/bench_gen.jsonl | cut -c1-400
