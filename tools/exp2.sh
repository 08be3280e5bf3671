#!/bin/bash
for ko in "" "d,c:4,b,c"; do
  echo "## KORDER=$ko"
  CUTENSOR_AMD_ABLATION=1 CUTENSOR_AMD_KORDER="$ko" python tools/tune_gett.py --problem einsum --splits 256 --kernels 54,66,67 --max 8 --reps 100 2>&1 | grep rank
done
for a in 54 66 67; do CUTENSOR_AMD_ABLATION=1 CUTENSOR_AMD_FORCE=$a:256 python tools/phase_timing.py 2>&1 | grep plan; done
CUTENSOR_AMD_KORDER="d,c:4,b,c" CUTENSOR_AMD_FORCE=54:256 python tools/phase_timing.py 2>&1 | grep plan
