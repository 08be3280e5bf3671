#!/bin/bash
# Round 6, GPU call u: the persistent kernel after the removal of its measurement instantiations and with the fp16 beta sum held in fp32:
# parity (bits of the one-tile twin), kernel traces of beta = 0.5 / 0 at 8192^3 and 8192^2 x 1024.
set -u
OUT=gpurun_out/r06u; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_h16p.py tests/test_gpu_h16.py tests/test_gpu_h16_unaligned.py -x -q > $OUT/h16.log 2>&1; echo "h16 rc $?"; tail -4 $OUT/h16.log
for beta in 0.5 0.0; do
  bash tools/trace_one.sh python $PWD/tools/h16_shape_sweep.py --layout mk,kn --beta $beta --only "8192,8192,8192;8192,8192,1024" > $OUT/trace_beta_$beta.txt 2>&1
  cat $OUT/trace_beta_$beta.txt | cut -c1-260
done
