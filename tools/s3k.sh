#!/bin/bash
O=gpurun_out/${1:-s3k}; mkdir -p $O
timeout 300 python tools/tune_gett.py --problem contraction --max 6 --reps 20 > $O/tune_contraction.jsonl 2>&1
timeout 300 python tools/tune_gett.py --problem gemm4096 --max 8 --reps 20 > $O/tune_gemm4096.jsonl 2>&1
timeout 100 python tools/phase_timing.py --dump $O/t.npy | grep plan > $O/phase.jsonl
