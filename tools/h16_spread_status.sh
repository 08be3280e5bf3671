#!/bin/bash
# Probe: the four-wave streamed bf16 kernel with its LDS-DMA pieces spread one per four MFMAs over a whole tile time
# (CUTENSOR_AMD_H16_SPREAD=1) beside one per two MFMAs in half the tile, the two-buffer four-wave kernel and the default.
# usage: tools/h16_spread_status.sh
cd ${GRAFT_REPO_ROOT:-.}
run() {  # run <waves> <spread> <layout> [--zeros]
  CUTENSOR_AMD_H16_WAVES=$1 CUTENSOR_AMD_H16_STAGES=4 CUTENSOR_AMD_H16_SPREAD=$2 timeout 120 python tools/bench_h16.py --layout $3 $4 2>&1 | grep workload | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves $1 spread $2 layout $3 $4: %.4f ms %.0f TF (%s)' % (d['ms_per_call'], d['tflops'], d['plan']['kname']))"
}
for L in km,kn mk,nk; do
  run 8 0 $L; run 4 0 $L; run 4s 0 $L; run 4s 1 $L
  run 8 0 $L --zeros; run 4 0 $L --zeros; run 4s 0 $L --zeros; run 4s 1 $L --zeros
done
