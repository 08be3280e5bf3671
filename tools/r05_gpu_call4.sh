#!/bin/bash
# Round 5, fourth GPU call (research build): the persistent kernel after the spill fix — timeline (EP 2) and the rate beside gett_h16w4x_kernel.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05d
mkdir -p $OUT
cd $ROOT
: > $OUT/h16p_timeline.jsonl
for shape in "4096 4096 4096" "8192 8192 512" "8192 8192 8192"; do
  timeout 120 python tools/h16p_timeline.py $shape 2>&1 | tail -1 >> $OUT/h16p_timeline.jsonl
  timeout 120 python tools/h16p_timeline.py $shape --zeros 2>&1 | tail -1 >> $OUT/h16p_timeline.jsonl
done
: > $OUT/h16p_vs_4x.jsonl
SH="8192,8192,8192;8192,8192,512;4096,4096,4096;8192,8192,1024;8192,8192,2048;4096,4096,8192"
for W in 4x 4p 4x 4p; do
  CUTENSOR_AMD_H16_WAVES=$W timeout 300 python tools/h16_shape_sweep.py --layout mk,kn --only "$SH" --reps 40 2>/dev/null >> $OUT/h16p_vs_4x.jsonl
done
cat $OUT/h16p_timeline.jsonl | cut -c1-600
cat $OUT/h16p_vs_4x.jsonl
