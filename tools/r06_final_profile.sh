#!/bin/bash
# Round 6: the rocprofv3 evidence of the final tree in one GPU call (kernel-trace runs and --pmc runs are separate invocations).
# usage: tools/r06_final_profile.sh <tag>  -> gpurun_out/<tag>/*.summary.txt, bench logs
set -u
TAG=${1:-r06_final}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
summ() { f=$(find $OUT/$1 -name '*.db' | head -1); [ -n "$f" ] && (cd $ROOT && python tools/rocprof_summary.py $f > $OUT/$2.summary.txt 2>&1); }
prof4() {   # prof4 <name> <cmd...>: trace, SQ, FETCH, WRITE
  n=$1; shift
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $OUT/${n}_trace -o r -- "$@" > $OUT/${n}_trace.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/${n}_sq -o r -- "$@" > $OUT/${n}_sq.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/${n}_fetch -o r -- "$@" > $OUT/${n}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/${n}_write -o r -- "$@" > $OUT/${n}_write.log 2>&1
  for p in trace sq fetch write; do summ ${n}_$p ${n}_$p; done
}
# 1. headline einsum: trace + three PMC passes (the recipe of rounds 2-5)
cd $ROOT && bash tools/gpu_profile.sh $TAG/einsum > $OUT/einsum_profile.log 2>&1
for d in trace pmc_sq pmc_fetch pmc_write; do cp $OUT/einsum/$d.summary.txt $OUT/einsum_$d.summary.txt 2>/dev/null; done
cp $OUT/einsum/pmc_traffic_einsum.json $OUT/pmc_traffic_einsum.json 2>/dev/null
# 2. bf16 8192^3 on the persistent kernel, beta = 0 and beta = 0.5
prof4 h16p_8192 python $ROOT/tools/h16_shape_sweep.py --layout mk,kn --only 8192,8192,8192 --reps 30
prof4 h16p_8192_beta python $ROOT/tools/h16_shape_sweep.py --layout mk,kn --only 8192,8192,8192 --reps 30 --beta 0.5
# 3. short contracted range: 16384^2 x 128 (two K-tiles per tile, staggered start)
prof4 h16p_short_k python $ROOT/tools/h16_shape_sweep.py --layout mk,kn --only 16384,16384,128 --reps 50
# 4. no 16-byte lanes: bf16 4100^3 (interior + strip launch)
prof4 h16_4100 python $ROOT/tools/h16_shape_sweep.py --layout mk,kn --only 4100,4100,4100 --reps 50
# 5. the whole default bench line: kernel trace only
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/bench_all_trace -o r -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu > $OUT/bench_all_trace.log 2>&1
summ bench_all_trace bench_all_trace
# 6. un-profiled reference lines
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2>&1
python bench.py > $OUT/bench_default.log 2>&1
find $OUT -name '*.csv' -size +1M -delete
find $OUT -name '*.db' -delete
rm -rf $OUT/einsum/trace $OUT/einsum/pmc_sq $OUT/einsum/pmc_fetch $OUT/einsum/pmc_write
du -sh $OUT; ls $OUT | head -60
