#!/bin/bash
# Round 5, fifth GPU call (research build): the STREAMING persistent kernel — parity, timeline, rate beside gett_h16w4x_kernel.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05e
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_h16p.py -x -q > $OUT/pytest_h16p.log 2>&1
echo "pytest h16p rc $?" >> $OUT/pytest_h16p.log
tail -25 $OUT/pytest_h16p.log | cut -c1-300
: > $OUT/h16p_timeline.jsonl
for shape in "4096 4096 4096" "8192 8192 512" "8192 8192 8192"; do
  timeout 120 python tools/h16p_timeline.py $shape 2>&1 | tail -1 >> $OUT/h16p_timeline.jsonl
  timeout 120 python tools/h16p_timeline.py $shape --zeros 2>&1 | tail -1 >> $OUT/h16p_timeline.jsonl
done
: > $OUT/h16p_vs_4x.jsonl
SH="8192,8192,8192;8192,8192,512;4096,4096,4096;8192,8192,1024;8192,8192,2048;4096,4096,8192;8192,8192,4096"
for W in 4x 4p 4x 4p; do
  CUTENSOR_AMD_H16_WAVES=$W timeout 300 python tools/h16_shape_sweep.py --layout mk,kn --only "$SH" --reps 40 2>/dev/null >> $OUT/h16p_vs_4x.jsonl
done
for W in 4x 4p; do
  CUTENSOR_AMD_H16_WAVES=$W timeout 300 python tools/h16_shape_sweep.py --layout km,kn --only "$SH" --reps 40 2>/dev/null >> $OUT/h16p_vs_4x.jsonl
done
cat $OUT/h16p_timeline.jsonl | cut -c1-700
cat $OUT/h16p_vs_4x.jsonl
