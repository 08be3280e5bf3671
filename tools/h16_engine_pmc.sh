#!/bin/bash
# Instruction mix / LDS activity of the engine's 16-bit kernel (whatever CUTENSOR_AMD_H16_WAVES selects; default: gett_h16w4x_kernel),
# 8192^3 bf16: two rocprofv3 --pmc passes (never combined with tracing).  usage: tools/h16_engine_pmc.sh <tag> [layout]
TAG=${1:-h16epmc}; LAY=${2:-km,kn}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
P1="SQ_WAVES SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH"
E="python $ROOT/tools/bench_h16.py --layout $LAY --reps 20"
rocprofv3 --pmc $P1 -d $OUT/e1 -o r -- $E > $OUT/e1.log 2>&1
rocprofv3 --pmc $P2 -d $OUT/e2 -o r -- $E > $OUT/e2.log 2>&1
cd $ROOT
python - $OUT <<'PY' > $OUT/summary.txt
import sqlite3, sys, glob, os
out = sys.argv[1]
for d in ("e1", "e2"):
    for db in glob.glob(os.path.join(out, d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        try:
            rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name having count(*) >= 15"))
        except sqlite3.Error as e:
            print("#", d, e); continue
        for k, n, v, cnt in rows:
            if "elementwise" in k or "distribution" in k: continue
            print("%s | %-60s | %-32s | %.6g | n=%d" % (d, k[:60], n, v, cnt))
PY
find $OUT -name '*.db' -delete; find $OUT -name '*.csv' -delete
cat $OUT/summary.txt
