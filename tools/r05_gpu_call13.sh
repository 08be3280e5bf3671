#!/bin/bash
# Round 5, GPU call 13: the round's rocprofv3 evidence on the production build (kernel-trace runs and --pmc runs are separate
# invocations) and the driver-style bench lines.  Trimmed form of tools/gpu_profile_all.sh: headline, bf16 8192^3 on the persistent
# kernel, the ragged-K shape, the whole bench line.
set -u
TAG=r05z
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
summ() { f=$(find $OUT/$1 -name '*.db' | head -1); [ -n "$f" ] && (cd $ROOT && python tools/rocprof_summary.py $f > $OUT/$2.summary.txt 2>&1); }
cd $ROOT && bash tools/gpu_profile.sh $TAG/einsum > $OUT/einsum_profile.log 2>&1
for d in trace pmc_sq pmc_fetch pmc_write; do cp $OUT/einsum/$d.summary.txt $OUT/einsum_$d.summary.txt 2>/dev/null; done
cp $OUT/einsum/pmc_traffic_einsum.json $OUT/pmc_traffic_einsum.json 2>/dev/null
cd /tmp
H16="python $ROOT/tools/bench_h16.py --layout mk,kn --reps 30"
rocprofv3 --kernel-trace --stats -d $OUT/h16_trace -o r -- $H16 > $OUT/h16_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/h16_sq -o r -- $H16 > $OUT/h16_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/h16_fetch -o r -- $H16 > $OUT/h16_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/h16_write -o r -- $H16 > $OUT/h16_write.log 2>&1
for p in trace sq fetch write; do summ h16_$p h16p_8192_mk_kn_$p; done
R="python $ROOT/tools/h16_shape_sweep.py --only 4096,4096,4104 --reps 200"
rocprofv3 --kernel-trace --stats -d $OUT/rag_trace -o r -- $R > $OUT/rag_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/rag_sq -o r -- $R > $OUT/rag_sq.log 2>&1
for p in trace sq; do summ rag_$p h16_ragged_4096_4104_$p; done
rocprofv3 --kernel-trace --stats -d $OUT/bench_all_trace -o r -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu --no-pmc > $OUT/bench_all_trace.log 2>&1
summ bench_all_trace bench_all_trace
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2> $OUT/bench_steps20.err; echo "bench steps20 rc $? lines $(wc -l < $OUT/bench_steps20.log)"
python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench default rc $? lines $(wc -l < $OUT/bench_default.log)"
find $OUT -name '*.csv' -size +1M -delete
find $OUT -name '*.db' -delete
rm -rf $OUT/einsum/trace $OUT/einsum/pmc_sq $OUT/einsum/pmc_fetch $OUT/einsum/pmc_write
du -sh $OUT; ls $OUT | head -40
