#!/bin/bash
# Round-3 closing profile of the default 16-bit kernel (gett_h16w4x_kernel): all four layouts beside the vendor GEMM (yardstick only) and
# the MFMA-only rates of the box in ONE call, a kernel trace and the HBM traffic counters (separate --pmc passes).
# usage: tools/h16_final_profile.sh <tag>
TAG=${1:-h16final}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
: > $OUT/h16_vs_vendor.jsonl
for L in mk,kn km,kn mk,nk km,nk; do
  python tools/bench_h16.py --layout $L 2>/dev/null | grep workload >> $OUT/h16_vs_vendor.jsonl
done
python tools/bench_h16.py --layout km,kn --zeros 2>/dev/null | grep workload >> $OUT/h16_vs_vendor.jsonl
python tools/ubench/vendor_gemm_bf16.py 2>/dev/null | grep vendor >> $OUT/h16_vs_vendor.jsonl
python - >> $OUT/h16_vs_vendor.jsonl 2>/dev/null <<PY
import ctypes, json, sys
sys.path.insert(0, '.')
from cudalibrarysamples_amd import cutensor as ct
import torch
torch.cuda.init()
out = {}
for name, kind, shape in (('zeros_16x16x32', 0, 1), ('uniform_16x16x32', 1, 1), ('uniform_32x32x16', 1, 0)):
    v = ctypes.c_float(0); ct.lib.ctamdMeasureMfmaCeilingShape(1, kind, shape, ctypes.byref(v)); out[name] = v.value
print(json.dumps({'mfma_only_tflops': out}))
PY
cd /tmp
H16="python $ROOT/tools/bench_h16.py --layout km,kn --reps 30"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- $H16 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o r -- $H16 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/write -o r -- $H16 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/sq -o r -- $H16 > $OUT/sq.log 2>&1
cd $ROOT
python - $OUT <<'PY' > $OUT/summary.txt
import sqlite3, sys, glob, os
out = sys.argv[1]
for db in glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True):
    c = sqlite3.connect(db)
    for name, n, avg, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name order by sum(duration) desc limit 4"):
        print("trace | %-90s | calls %d | avg_us %.2f | min_us %.2f" % (name[:90], n, avg / 1e3, mn / 1e3))
for d in ("fetch", "write", "sq"):
    for db in glob.glob(os.path.join(out, d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        try:
            rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name having count(*) >= 15"))
        except sqlite3.Error as e:
            print("#", d, e); continue
        for k, n, v, cnt in rows:
            if "gett_h16" not in k: continue
            print("%s | %-60s | %-28s | %.6g | n=%d" % (d, k[:60], n, v, cnt))
PY
find $OUT -name '*.db' -delete; find $OUT -name '*.csv' -delete
cat $OUT/h16_vs_vendor.jsonl | cut -c1-400; cat $OUT/summary.txt
