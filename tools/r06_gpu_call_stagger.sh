#!/bin/bash
# Round 6: staggered start of the persistent kernel's workgroups on short contracted ranges (CUTENSOR_AMD_H16P_STAGGER = phase step in
# units of 1024 cycles; workgroup w of an XCD starts ((w / 8) % 4) steps late), alternating with the default on one box.
set -u
OUT=gpurun_out/r06zh; mkdir -p $OUT
export TMPDIR=/tmp CTAMD_LIB_FLAVOUR=hooks
SH="16384,16384,128;8192,8192,128;8192,8192,256;8192,8192,512;8192,8192,1024;8192,8192,2048;4096,4096,128;16384,4096,256"
for rep in 1 2; do
  for st in 0 -1; do
    CUTENSOR_AMD_H16P_STAGGER=$st timeout 200 python tools/h16_shape_sweep.py --layout km,kn --only "$SH" 2>/dev/null | sed "s/^{/{\"stagger\": $st, /" >> $OUT/stagger.jsonl
  done
done
python - <<'PY'
import json, collections
r = collections.defaultdict(list)
for l in open("gpurun_out/r06zh/stagger.jsonl"):
    d = json.loads(l); r[(d["M"], d["K"], d["stagger"])].append(round(d["ms"] * 1e3, 1))
for k in sorted(r): print(k, r[k])
PY
unset CTAMD_LIB_FLAVOUR
timeout 900 python -m pytest tests/test_gpu_h16p.py -x -q > $OUT/h16p.log 2>&1; tail -2 $OUT/h16p.log
timeout 600 python tools/bench_einsum_shapes.py > $OUT/einsum_shapes.jsonl 2>/dev/null; python - <<'PY'
import json
for l in open("gpurun_out/r06zh/einsum_shapes.jsonl"):
    d = json.loads(l); print(d["equation"], d["extents"], d["us"], "vendor", d["vendor_us"], d["kernel"])
PY
