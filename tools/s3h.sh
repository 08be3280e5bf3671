#!/bin/bash
O=gpurun_out/${1:-s3h}; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_h16.py -x -q ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for l in mk,kn km,kn mk,nk km,nk; do timeout 120 python tools/bench_h16.py --layout $l 2>&1 | grep workload; done > $O/bench_h16.jsonl
cat $O/bench_h16.jsonl | cut -c1-400
