#!/bin/bash
# Round 5, third GPU call: the persistent kernel's epilogue variants (EP 0 / 1 / 2) under in-kernel stamps, then EP 2 beside
# gett_h16w4x_kernel on the shapes of the second call, then its parity file.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05c
mkdir -p $OUT
cd $ROOT
: > $OUT/h16p_epilogue_variants.jsonl
for ep in 0 1 2; do
  for shape in "4096 4096 4096" "8192 8192 512" "8192 8192 8192"; do
    CUTENSOR_AMD_H16P_EP=$ep timeout 120 python tools/h16p_timeline.py $shape 2>&1 | tail -1 >> $OUT/h16p_epilogue_variants.jsonl
    CUTENSOR_AMD_H16P_EP=$ep timeout 120 python tools/h16p_timeline.py $shape --zeros 2>&1 | tail -1 >> $OUT/h16p_epilogue_variants.jsonl
  done
done
cut -c1-700 $OUT/h16p_epilogue_variants.jsonl
timeout 600 python -m pytest tests/test_gpu_h16p.py -x -q > $OUT/pytest_h16p.log 2>&1
echo "pytest h16p rc $?" >> $OUT/pytest_h16p.log
tail -5 $OUT/pytest_h16p.log
: > $OUT/h16p_vs_4x.jsonl
SH="8192,8192,8192;8192,8192,512;4096,4096,4096;8192,8192,1024;8192,8192,2048;4096,4096,8192"
for W in 4x 4p 4x 4p; do
  for L in mk,kn km,kn; do
    CUTENSOR_AMD_H16_WAVES=$W timeout 300 python tools/h16_shape_sweep.py --layout $L --only "$SH" --reps 40 2>/dev/null >> $OUT/h16p_vs_4x.jsonl
  done
done
cat $OUT/h16p_vs_4x.jsonl
