#!/bin/bash
# Round 6, end: the whole GPU suite on the final tree, rocprofv3 traces of the new paths (block permutation; a repacked bf16 / fp32
# contraction; the sweep mask), the headline profile and the driver-style bench lines.
set -u
TAG=${1:-r06zzg}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_gpu.log
summ() { f=$(find $OUT/$1 -name '*.db' | head -1); [ -n "$f" ] && (cd $ROOT && python tools/rocprof_summary.py $f > $OUT/$2.summary.txt 2>&1); }
trace() { n=$1; shift; cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/${n}_trace -o r -- "$@" > $OUT/${n}_trace.log 2>&1; summ ${n}_trace ${n}_trace; cd $ROOT; }
pmc() { n=$1; shift; cd /tmp
  rocprofv3 --pmc FETCH_SIZE -d $OUT/${n}_fetch -o r -- "$@" > $OUT/${n}_fetch.log 2>&1; summ ${n}_fetch ${n}_fetch
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/${n}_write -o r -- "$@" > $OUT/${n}_write.log 2>&1; summ ${n}_write ${n}_write; cd $ROOT; }
trace block_permute python $ROOT/tools/bench_block_permute.py
pmc block_permute python $ROOT/tools/bench_block_permute.py
EINSUM_SHAPES_SET=sweep trace sweep_shapes_bf16 python $ROOT/tools/bench_einsum_shapes.py bf16
EINSUM_SHAPES_SET=sweep trace sweep_shapes_f32 python $ROOT/tools/bench_einsum_shapes.py f32
bash tools/gpu_profile.sh $TAG/einsum > $OUT/einsum_profile.log 2>&1
for d in trace pmc_sq pmc_fetch pmc_write; do cp $OUT/einsum/$d.summary.txt $OUT/einsum_$d.summary.txt 2>/dev/null; done
cp $OUT/einsum/pmc_traffic_einsum.json $OUT/pmc_traffic_einsum.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2>&1; tail -1 $OUT/bench_steps20.log | cut -c1-400
python bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log | cut -c1-400
find $OUT -name '*.csv' -size +1M -delete
find $OUT -name '*.db' -delete
rm -rf $OUT/einsum/trace $OUT/einsum/pmc_sq $OUT/einsum/pmc_fetch $OUT/einsum/pmc_write
du -sh $OUT
