#!/bin/bash
# Round 5, GPU call 14: MFMA issue order experiment (build/exp_snake/libcutensor.so: -DCTAMD_MFMA_SNAKE) against the production library,
# alternating runs on one box; parity of the variant first.
set -u
OUT=gpurun_out/r05o; mkdir -p $OUT
export TMPDIR=/tmp
SN=$PWD/build/exp_snake/libcutensor.so
CUTENSOR_AMD_LIBRARY=$SN timeout 600 python -m pytest tests/test_gpu_h16p.py tests/test_gpu_h16.py -x -q -k "persistent or gemm_like or full_size or ragged_k_stays" > $OUT/pytest_snake.log 2>&1; echo "snake parity rc $?"; tail -2 $OUT/pytest_snake.log
for rep in 1 2 3; do
  for lay in mk,kn km,kn; do
    timeout 120 python tools/bench_h16.py --layout $lay 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib':'production','layout':'$lay','tflops':round(d['tflops'],1),'kernel_tflops':round(d['kernel_tflops'],1),'kname':d['plan']['kname']}))" >> $OUT/snake_ab.jsonl
    CUTENSOR_AMD_LIBRARY=$SN timeout 120 python tools/bench_h16.py --layout $lay 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib':'snake','layout':'$lay','tflops':round(d['tflops'],1),'kernel_tflops':round(d['kernel_tflops'],1),'kname':d['plan']['kname']}))" >> $OUT/snake_ab.jsonl
  done
done
timeout 120 python tools/bench_h16.py --zeros 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib':'production','zeros':1,'tflops':round(d['tflops'],1)}))" >> $OUT/snake_ab.jsonl
CUTENSOR_AMD_LIBRARY=$SN timeout 120 python tools/bench_h16.py --zeros 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib':'snake','zeros':1,'tflops':round(d['tflops'],1)}))" >> $OUT/snake_ab.jsonl
CUTENSOR_AMD_LIBRARY=$SN timeout 200 python tools/h16_shape_sweep.py --only "4096,4096,4096;2048,2048,2048;8192,8192,1024" > $OUT/snake_sweep.jsonl 2>/dev/null
timeout 200 python tools/h16_shape_sweep.py --only "4096,4096,4096;2048,2048,2048;8192,8192,1024" > $OUT/prod_sweep.jsonl 2>/dev/null
cat $OUT/snake_ab.jsonl; echo; cat $OUT/snake_sweep.jsonl $OUT/prod_sweep.jsonl
