#!/bin/bash
# Round 5, GPU call 16: MFMA order, second variant (B fragment stationary: build/exp_col/libcutensor.so, -DCTAMD_MFMA_COLSNAKE) against
# the production library (A fragment stationary, even rows reversed), alternating runs on one box.
set -u
OUT=gpurun_out/r05q; mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/build/exp_col/libcutensor.so
CUTENSOR_AMD_LIBRARY=$V timeout 600 python -m pytest tests/test_gpu_h16p.py tests/test_gpu_h16.py -x -q -k "persistent or gemm_like or full_size" > $OUT/pytest_variant.log 2>&1; echo "variant parity rc $?"; tail -1 $OUT/pytest_variant.log
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib':'$1','layout':'$2','tflops':round(d['tflops'],1),'kernel_tflops':round(d['kernel_tflops'],1)}))"; }
for rep in 1 2 3; do
  for lay in mk,kn km,kn mk,nk; do
    timeout 120 python tools/bench_h16.py --layout $lay 2>/dev/null | one production $lay >> $OUT/order_ab.jsonl
    CUTENSOR_AMD_LIBRARY=$V timeout 120 python tools/bench_h16.py --layout $lay 2>/dev/null | one b_stationary $lay >> $OUT/order_ab.jsonl
  done
done
cat $OUT/order_ab.jsonl
