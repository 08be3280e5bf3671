set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06zv; mkdir -p $OUT; export TMPDIR=/tmp CTAMD_LIB_FLAVOUR=hooks
cd /tmp
rocprofv3 -L 2>/dev/null | grep -io "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u > $OUT/counters_avail.txt
wc -l $OUT/counters_avail.txt; grep -i "ICACHE\|IFETCH" $OUT/counters_avail.txt | head -20
SH="16384,16384,128;8192,8192,1024"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/icache -o r -- python $ROOT/tools/f32_shape.py "$SH" 10 > $OUT/icache.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/wait -o r -- python $ROOT/tools/f32_shape.py "$SH" 10 > $OUT/wait.log 2>&1
cd $ROOT
for n in icache wait; do f=$(find $OUT/$n -name '*.db' | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/$n.summary.txt 2>&1; tail -3 $OUT/$n.log; grep PMC $OUT/$n.summary.txt | grep -v fill | head -20; done
find $OUT -name '*.db' -delete; find $OUT -name '*.csv' -size +1M -delete
