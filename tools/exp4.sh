#!/bin/bash
# ring depth experiment on the b=48 variant of the headline einsum (24 K-tiles per slice)
echo "## einsum48 S sweep"
CUTENSOR_AMD_ABLATION=1 python tools/tune_gett.py --problem einsum48 --splits 256 --kernels 54,55,56,68,69 --max 8 --reps 100 2>&1 | grep rank
echo "## einsum48 S sweep KORDER d,c:4,b,c"
CUTENSOR_AMD_KORDER="d,c:4,b,c" python tools/tune_gett.py --problem einsum48 --splits 256 --kernels 54,55,56 --max 8 --reps 100 2>&1 | grep rank
for a in 54 55 56; do CUTENSOR_AMD_FORCE=$a:256 python tools/phase_timing.py --b 48 2>&1 | grep plan; done
CUTENSOR_AMD_FORCE=54:256 python tools/phase_timing.py 2>&1 | grep plan
