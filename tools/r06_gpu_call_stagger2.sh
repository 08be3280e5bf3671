#!/bin/bash
# Round 6: staggered first round of the ONE-TILE 16-bit kernel (odd K-tile counts, where the persistent kernel does not stream), on / off.
set -u
OUT=gpurun_out/r06zi; mkdir -p $OUT
export TMPDIR=/tmp CTAMD_LIB_FLAVOUR=hooks
SH="8192,8192,64;8192,8192,192;8192,8192,320;16384,16384,64;8192,8192,448"
for rep in 1 2; do
  for st in 0 -1 3 6; do
    CUTENSOR_AMD_H16P_STAGGER=$st timeout 200 python tools/h16_shape_sweep.py --layout mk,kn --only "$SH" 2>/dev/null | sed "s/^{/{\"stagger\": $st, /" >> $OUT/stagger_one_tile.jsonl
  done
done
python - <<'PY'
import json, collections
r = collections.defaultdict(list)
for l in open("gpurun_out/r06zi/stagger_one_tile.jsonl"):
    d = json.loads(l); r[(d["M"], d["K"], d["stagger"], d["kname"])].append(round(d["ms"] * 1e3, 1))
for k in sorted(r): print(k, r[k])
PY
unset CTAMD_LIB_FLAVOUR
timeout 900 python -m pytest tests/test_gpu_h16.py tests/test_gpu_h16_unaligned.py -x -q > $OUT/h16.log 2>&1; tail -2 $OUT/h16.log
timeout 600 python tools/bench_einsum_shapes.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['equation'], d['extents'], d['us'], 'vendor', d['vendor_us'], d['kernel'])"
