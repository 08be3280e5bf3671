#!/bin/bash
echo "## einsum"
CUTENSOR_AMD_ABLATION=1 python tools/tune_gett.py --problem einsum --splits 256 --kernels 54,70 --max 8 --reps 100 2>&1 | grep rank
echo "## einsum48 S sweep"
python tools/tune_gett.py --problem einsum48 --splits 256 --kernels 54,55,56 --max 8 --reps 100 2>&1 | grep rank
for a in 54 70; do CUTENSOR_AMD_ABLATION=1 CUTENSOR_AMD_FORCE=$a:256 python tools/phase_timing.py 2>&1 | grep plan; done
for a in 54 55 56; do CUTENSOR_AMD_FORCE=$a:256 python tools/phase_timing.py --b 48 2>&1 | grep plan; done
