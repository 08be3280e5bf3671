#!/bin/bash
# Round 5, GPU call 7: ragged K on the LDS-DMA 16-bit kernels (RAG instantiations) — parity, fuzz, rates; bench line last-line check.
set -u
OUT=gpurun_out/r05g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_h16.py tests/test_gpu_gen.py tests/test_gpu_h16p.py -x -q > $OUT/pytest_h16.log 2>&1; echo "pytest rc $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_h16.log
timeout 600 python tools/fuzz_contraction.py --ragged-k --cases 250 --seed 5 > $OUT/fuzz_ragged.log 2>&1; echo "fuzz ragged rc $?" | tee -a $OUT/summary.txt; tail -2 $OUT/fuzz_ragged.log
timeout 600 python tools/fuzz_contraction.py --ragged-k --strided --cases 150 --seed 6 > $OUT/fuzz_ragged_strided.log 2>&1; echo "fuzz ragged strided rc $?" | tee -a $OUT/summary.txt; tail -2 $OUT/fuzz_ragged_strided.log
timeout 300 python tools/fuzz_contraction.py --aligned --cases 150 --seed 7 > $OUT/fuzz_aligned.log 2>&1; echo "fuzz aligned rc $?" | tee -a $OUT/summary.txt; tail -1 $OUT/fuzz_aligned.log
for lay in mk,kn km,kn mk,nk km,nk; do
  timeout 300 python tools/h16_shape_sweep.py --layout $lay --only "4096,4096,4104;4096,4096,4096;8192,8192,8200;2048,2048,2056;2048,2048,2048;1024,1024,1032;1024,1024,1024" >> $OUT/ragged_rates.jsonl 2>$OUT/sweep.err
done
CUTENSOR_AMD_GEN=f timeout 300 python tools/h16_shape_sweep.py --only "4096,4096,4104;2048,2048,2056;1024,1024,1032" > $OUT/ragged_rates_gen_family.jsonl 2>>$OUT/sweep.err
cat $OUT/ragged_rates.jsonl $OUT/ragged_rates_gen_family.jsonl
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?" | tee -a $OUT/summary.txt
tail -n 1 $OUT/bench.out | cut -c1-300; echo; echo "bench stdout lines: $(wc -l < $OUT/bench.out)" | tee -a $OUT/summary.txt
