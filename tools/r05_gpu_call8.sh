#!/bin/bash
# Round 5, GPU call 8: ragged-K rates after the clock ramp (+ the vendor GEMM on the same shapes), the cuTENSORMg protocol decomposition,
# bf16 8192^3 on four layouts against the vendor (random data and zeros).
set -u
OUT=gpurun_out/r05h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_h16.py -x -q > $OUT/pytest_h16.log 2>&1; echo "pytest h16 rc $?" | tee -a $OUT/summary.txt; tail -2 $OUT/pytest_h16.log
SH="4096,4096,4096;4096,4096,4104;4096,4096,4160;8192,8192,8200;2048,2048,2056;1024,1024,1032"
for lay in mk,kn km,kn mk,nk km,nk; do
  timeout 300 python tools/h16_shape_sweep.py --layout $lay --only "$SH" >> $OUT/ragged_rates.jsonl 2>$OUT/sweep.err
done
CUTENSOR_AMD_GEN=f timeout 300 python tools/h16_shape_sweep.py --only "4096,4096,4096;4096,4096,4104;2048,2048,2056;1024,1024,1032" > $OUT/ragged_rates_gen_family.jsonl 2>>$OUT/sweep.err
timeout 300 python tools/ubench/vendor_gemm_bf16.py --shapes "$SH" > $OUT/ragged_vendor.jsonl 2>>$OUT/sweep.err
cat $OUT/ragged_rates.jsonl $OUT/ragged_rates_gen_family.jsonl $OUT/ragged_vendor.jsonl
timeout 300 python tools/mg_protocol_decompose.py 4096 > $OUT/mg_protocol.json 2> $OUT/mg_protocol.err; cat $OUT/mg_protocol.json; tail -2 $OUT/mg_protocol.err
for i in 1 2 3; do timeout 120 oracle/_ref/contraction_multi_gpu 2>&1 | grep -i "took\|success\|error" ; done | tee $OUT/ref_sample_mg.txt
for lay in mk,kn km,kn mk,nk km,nk; do
  timeout 200 python tools/bench_h16.py --layout $lay >> $OUT/h16_8192_engine.jsonl 2>>$OUT/sweep.err
done
timeout 200 python tools/bench_h16.py --zeros >> $OUT/h16_8192_engine.jsonl 2>>$OUT/sweep.err
timeout 200 python tools/ubench/vendor_gemm_bf16.py > $OUT/h16_8192_vendor.jsonl 2>>$OUT/sweep.err
timeout 200 python tools/ubench/vendor_gemm_bf16.py --zeros >> $OUT/h16_8192_vendor.jsonl 2>>$OUT/sweep.err
cat $OUT/h16_8192_engine.jsonl $OUT/h16_8192_vendor.jsonl
