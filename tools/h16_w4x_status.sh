#!/bin/bash
# The 16x16x32 four-wave bf16 kernel (CUTENSOR_AMD_H16_WAVES=4x) beside the eight-wave kernel and the lean 32x32x16 four-wave kernel
# (=4v), U(-1,1) and zero-filled operands, all four layouts.  usage: tools/h16_w4x_status.sh
cd ${GRAFT_REPO_ROOT:-.}
run() {  # run <waves> <layout> [--zeros]
  CUTENSOR_AMD_H16_WAVES=$1 timeout 120 python tools/bench_h16.py --layout $2 $3 2>&1 | grep workload | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves $1 layout $2 $3: %.4f ms %.0f TF (%s)' % (d['ms_per_call'], d['tflops'], d['plan']['kname']))"
}
for L in ${LAYOUTS:-km,kn mk,kn mk,nk km,nk}; do
  run 8 $L; run 4v $L; run 4x $L
  run 8 $L --zeros; run 4x $L --zeros
done
