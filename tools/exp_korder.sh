#!/bin/bash
# experiment: K-digit order x stream kernel (+ memory-only / compute-only ablations) on the headline einsum
for ko in "" "d,b,c" "d,c:4,b,c" "d,c:2,b,c" "d,c:8,b,c" "d,c:16,b,c"; do
  echo "## KORDER=$ko"
  CUTENSOR_AMD_ABLATION=1 CUTENSOR_AMD_KORDER="$ko" python tools/tune_gett.py --problem einsum --splits 256 --kernels 54,55,68,69 --max 8 2>&1 | grep rank
done
