#!/bin/bash
# Round 5, second GPU call: parity of the persistent 16-bit kernel, then its rate beside gett_h16w4x_kernel on the same box
# (zeros and U(-1,1), 8192^3 and the short-K / one-round shapes the vendor yardstick showed behind), then the GPU suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05b
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_h16p.py -x -q > $OUT/pytest_h16p.log 2>&1
echo "pytest h16p rc $?" >> $OUT/pytest_h16p.log
tail -30 $OUT/pytest_h16p.log
: > $OUT/h16p_vs_4x.jsonl
SH="8192,8192,8192;8192,8192,512;4096,4096,4096;8192,8192,1024;8192,8192,2048;4096,4096,8192"
for rep in 1 2; do
  for W in 4x 4p; do
    for L in mk,kn km,kn; do
      CUTENSOR_AMD_H16_WAVES=$W timeout 300 python tools/h16_shape_sweep.py --layout $L --only "$SH" --reps 40 2>/dev/null >> $OUT/h16p_vs_4x.jsonl
    done
  done
done
for W in 4x 4p; do
  CUTENSOR_AMD_H16_WAVES=$W timeout 120 python tools/bench_h16.py --zeros 2>/dev/null | grep workload | cut -c1-600 >> $OUT/h16p_vs_4x_zeros.jsonl
  CUTENSOR_AMD_H16_WAVES=$W timeout 120 python tools/bench_h16.py 2>/dev/null | grep workload | cut -c1-600 >> $OUT/h16p_vs_4x_zeros.jsonl
done
cat $OUT/h16p_vs_4x.jsonl
cat $OUT/h16p_vs_4x_zeros.jsonl
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_h16p.py > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log | cut -c1-300
