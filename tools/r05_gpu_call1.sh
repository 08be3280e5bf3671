#!/bin/bash
# Round 5, first GPU call: the GPU suite on the tree with complex reductions / the ABI guard, the tile-barrier decomposition of
# gett_h16w4x_kernel (zero-filled operands), and the same-box vendor yardstick at the mid-size bf16 shapes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05a
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
: > $OUT/w4x_barrier_decomposition.jsonl
for x in 0 8 9 10 0 8 9 10; do
  CUTENSOR_AMD_H16_XST=$x timeout 120 python tools/h16_wg_timeline.py --zeros 2>/dev/null | tail -1 >> $OUT/w4x_barrier_decomposition.jsonl
done
CUTENSOR_AMD_H16_XST=0 timeout 120 python tools/h16_wg_timeline.py 2>/dev/null | tail -1 >> $OUT/w4x_barrier_decomposition.jsonl
SH="2048,2048,2048;1024,1024,1024;4096,1024,4096;8192,8192,512;4096,4096,4096;2048,2048,16384"
timeout 300 python tools/ubench/vendor_gemm_bf16.py --shapes "$SH" --reps 50 > $OUT/h16_mid_vendor.jsonl 2>/dev/null
: > $OUT/h16_mid_engine.jsonl
for L in mk,kn km,kn mk,nk; do
  timeout 300 python tools/h16_shape_sweep.py --layout $L --only "$SH" --reps 50 2>/dev/null >> $OUT/h16_mid_engine.jsonl
done
tail -3 $OUT/pytest_gpu.log
cat $OUT/w4x_barrier_decomposition.jsonl | cut -c1-400
