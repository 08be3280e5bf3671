#!/bin/bash
# Round 6: odd K-tile counts in the persistent kernel (a zero K-tile appended): parity, then against the one-tile kernel on one box.
set -u
OUT=gpurun_out/r06zp; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_h16p.py tests/test_gpu_h16.py tests/test_gpu_h16_unaligned.py tests/test_gpu_einsum.py tests/test_gpu_torch_binding.py tests/test_gpu_ref_torch_binding.py -x -q > $OUT/h16.log 2>&1
grep -n "passed\|failed" $OUT/h16.log; grep -n "Error\|assert " $OUT/h16.log | head -4 | cut -c1-500
export CTAMD_LIB_FLAVOUR=hooks
timeout 400 python tools/fuzz_contraction.py --cases 400 --seed 91 --aligned > $OUT/fuzz_aligned.log 2>&1; tail -1 $OUT/fuzz_aligned.log | cut -c1-160
SH="8192,8192,64;8192,8192,192;8192,8192,320;8192,8192,448;8192,8192,960;16384,16384,64;8192,8192,128;8192,8192,8192"
for rep in 1 2; do
  timeout 300 python tools/h16_shape_sweep.py --layout mk,kn --only "$SH" 2>/dev/null | sed "s/^{/{\"lib\": \"default\", /" >> $OUT/odd_k_ab.jsonl
  CUTENSOR_AMD_H16P=0 timeout 300 python tools/h16_shape_sweep.py --layout mk,kn --only "$SH" 2>/dev/null | sed "s/^{/{\"lib\": \"one_tile\", /" >> $OUT/odd_k_ab.jsonl
done
python - <<'PY'
import json, collections
r = collections.defaultdict(list)
for l in open("gpurun_out/r06zp/odd_k_ab.jsonl"):
    d = json.loads(l); r[(d["M"], d["K"], d["lib"], d["kname"])].append(round(d["ms"] * 1e3, 1))
for k in sorted(r): print(k, r[k])
PY
unset CTAMD_LIB_FLAVOUR
timeout 600 python tools/bench_einsum_shapes.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['equation'], list(d['extents'].values()), d['us'], 'vendor', d['vendor_us'], d['kernel'], d['max_rel_diff_vs_vendor'])"
