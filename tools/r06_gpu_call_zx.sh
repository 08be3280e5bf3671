#!/bin/bash
# Round 6: fp32 tiles leave as whole rows through a per-wave LDS image (gett_store_tile_f32_rows) — parity suites, per-workgroup timeline
# on 16384^2 x 128, the einsum-library shapes in fp32 beside the vendor BLAS, the aligned shapes that must not lose.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r06zx}; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_contraction.py tests/test_gpu_einsum.py tests/test_gpu_f32_unaligned.py tests/test_gpu_torch_binding.py tests/test_gpu_trinary.py tests/test_gpu_lone_modes.py -x -q > $OUT/pytest_f32.log 2>&1; tail -3 $OUT/pytest_f32.log
export CTAMD_LIB_FLAVOUR=hooks
for a in "" 1; do F32_TIMELINE_ALGO=$a timeout 120 python tools/f32_wg_timeline.py 16384 16384 128 2>&1 | tail -1 >> $OUT/f32_timeline_flat128.jsonl; done
timeout 120 python tools/f32_wg_timeline.py 4096 4096 4096 2>&1 | tail -1 >> $OUT/f32_timeline_flat128.jsonl
timeout 600 python tools/bench_einsum_shapes.py f32 2>/dev/null > $OUT/einsum_shapes_f32.jsonl
timeout 300 python tools/f32_shape.py "4096,4096,4096;8192,8192,1024;8192,8192,256;2048,2048,2048;4098,4098,4098" 20 > $OUT/f32_shapes.jsonl 2>&1
python - <<PY
import json
for l in open("$OUT/f32_timeline_flat128.jsonl"):
    d = json.loads(l); c = d['cycles_mean']
    print(d['shape'], d['algo'], d['kname'][:26], d['tile'], 'pf', d['pf'], {k: round(v) for k, v in c.items()}, 'wg_us', round(d['wg_dur_us_mean'], 1), 'ms', round(d['ms_per_call'], 3))
for l in open("$OUT/einsum_shapes_f32.jsonl"):
    d = json.loads(l); print(d['equation'], d['extents'], d['us'], 'vendor', d['vendor_us'], d['kernel'], d['tile'], 'err', d['max_rel_diff_vs_vendor'])
for l in open("$OUT/f32_shapes.jsonl"):
    print(l.strip())
PY
