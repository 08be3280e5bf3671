#!/bin/bash
# The measurement instantiations of gett_h16w4x_kernel (CUTENSOR_AMD_H16_XST: 0 default, 4 early fragment reads, 5 staggered first
# round) on bf16 8192^3 mk,kn, zeros and U(-1,1) data, interleaved twice on one box.  usage: tools/h16_w4x_variants.sh OUT.jsonl
out=${1:-gpurun_out/h16_w4x_variants.jsonl}
: > "$out"
for rep in 1 2; do
  for x in 0 4 5; do
    for z in "--zeros" ""; do
      CUTENSOR_AMD_H16_XST=$x timeout 300 python tools/h16_wg_timeline.py $z >> "$out" 2>> "$out.err"
    done
  done
done
