#!/bin/bash
# Status of the four-wave streamed bf16 kernel (CUTENSOR_AMD_H16_WAVES=4s) beside the default, random and zero-filled operands,
# ring depths 4 and 5, all four layouts.  usage: tools/h16_w4s_status.sh
cd ${GRAFT_REPO_ROOT:-.}
run() {  # run <waves> <stages> <layout> [--zeros]
  CUTENSOR_AMD_H16_WAVES=$1 CUTENSOR_AMD_H16_STAGES=$2 timeout 120 python tools/bench_h16.py --layout $3 $4 2>&1 | grep workload | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves $1 stages $2 layout $3 $4: %.4f ms %.0f TF (%s)' % (d['ms_per_call'], d['tflops'], d['plan']['kname']))"
}
for L in km,kn mk,kn mk,nk km,nk; do
  run 8 5 $L; run 4r 5 $L; run 4s 4 $L
  run 8 5 $L --zeros; run 4r 5 $L --zeros; run 4s 4 $L --zeros
done
