#!/bin/bash
# Round 6, review item 8(b): the XCD tile-block shape of the persistent 16-bit kernel — 8 x 4 (shipped), 4 x 8, 16 x 2 (measurement builds
# -DCTAMD_P_XCD_GROUP=4 / 16 in build/exp_g4, build/exp_g16) — time, FETCH_SIZE and the sustained clock at bf16 8192^3; then the forced 16-bit
# kernel variants on the attention-score shapes with one / two K-tiles per tile.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06zt; mkdir -p $OUT
export TMPDIR=/tmp CTAMD_LIB_FLAVOUR=hooks
cd $ROOT
for rep in 1 2; do
  for g in 8 4 16; do
    LIB=$ROOT/cudalibrarysamples_amd/lib_hooks/libcutensor.so; [ $g != 8 ] && LIB=$ROOT/build/exp_g$g/libcutensor.so
    for lay in mk,kn km,kn mk,nk km,nk; do
      CUTENSOR_AMD_LIBRARY=$LIB timeout 200 python tools/h16_shape_sweep.py --layout $lay --only "8192,8192,8192;8192,8192,2048" --reps 30 2>/dev/null | sed "s/^{/{\"xcd_group\": $g, /" >> $OUT/xcd_group_times.jsonl
    done
  done
done
summ() { f=$(find $OUT/$1 -name '*.db' | head -1); [ -n "$f" ] && (cd $ROOT && python tools/rocprof_summary.py $f > $OUT/$1.summary.txt 2>&1); }
for g in 8 4 16; do
  LIB=$ROOT/cudalibrarysamples_amd/lib_hooks/libcutensor.so; [ $g != 8 ] && LIB=$ROOT/build/exp_g$g/libcutensor.so
  export CUTENSOR_AMD_LIBRARY=$LIB
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE -d $OUT/g${g}_fetch -o r -- python $ROOT/tools/h16_shape_sweep.py --layout mk,kn --only 8192,8192,8192 --reps 20 > $OUT/g${g}_fetch.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/g${g}_sq -o r -- python $ROOT/tools/h16_shape_sweep.py --layout mk,kn --only 8192,8192,8192 --reps 20 > $OUT/g${g}_sq.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/g${g}_trace -o r -- python $ROOT/tools/h16_shape_sweep.py --layout mk,kn --only 8192,8192,8192 --reps 20 > $OUT/g${g}_trace.log 2>&1
  summ g${g}_fetch; summ g${g}_sq; summ g${g}_trace
  unset CUTENSOR_AMD_LIBRARY
done
cd $ROOT
# attention-score shapes (cases 0 and 2) and the batched 4-K-tile case (7) under every kernel of the family
for rep in 1 2; do
  for w in planner 4x 4p 4m 4m4 4q; do
    if [ $w = planner ]; then unset CUTENSOR_AMD_H16_WAVES; else export CUTENSOR_AMD_H16_WAVES=$w; fi
    EINSUM_SHAPES_ONLY=0,2,7 timeout 300 python tools/bench_einsum_shapes.py 2>/dev/null | sed "s/^{/{\"forced\": \"$w\", /" >> $OUT/scores_forced_variants.jsonl
  done
done
unset CUTENSOR_AMD_H16_WAVES
python - <<'PY'
import json, collections
r = collections.defaultdict(list)
for l in open("gpurun_out/r06zt/xcd_group_times.jsonl"):
    d = json.loads(l); r[(d["K"], d["layout"], d["xcd_group"])].append(d["tflops"])
for k in sorted(r): print(k, r[k])
r = collections.defaultdict(list)
for l in open("gpurun_out/r06zt/scores_forced_variants.jsonl"):
    d = json.loads(l); r[(d["equation"], d["extents"]["d"] if "d" in d["extents"] else d["extents"]["k"], d["forced"], d["kernel"])].append((d["us"], d["vendor_us"]))
for k in sorted(r): print(k, r[k])
PY
find $OUT -name '*.csv' -size +1M -delete; find $OUT -name '*.db' -delete
for g in 8 4 16; do echo "== group $g"; grep -i "gett_h16w4p" $OUT/g${g}_fetch.summary.txt | head -3; grep -i "gett_h16w4p" $OUT/g${g}_sq.summary.txt | head -3; grep -i "gett_h16w4p" $OUT/g${g}_trace.summary.txt | head -2; done
