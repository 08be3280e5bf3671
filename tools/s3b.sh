#!/bin/bash
O=gpurun_out/s3b; mkdir -p $O
for a in 54 68 69 70; do CUTENSOR_AMD_ABLATION=1 CUTENSOR_AMD_FORCE=$a:256 python tools/phase_timing.py 2>&1 | grep plan; done > $O/phase.jsonl
