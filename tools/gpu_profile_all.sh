#!/bin/bash
# The round's rocprofv3 evidence in one GPU call.  Kernel-trace runs and --pmc runs are separate invocations (never combined).
# usage: tools/gpu_profile_all.sh <tag>      -> gpurun_out/<tag>/*.summary.txt (+ pmc_traffic_*.json, bench logs)
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
summ() {   # summ <dir> <name>
  f=$(find $OUT/$1 -name '*.db' | head -1)
  [ -n "$f" ] && (cd $ROOT && python tools/rocprof_summary.py $f > $OUT/$2.summary.txt 2>&1)
}
# ---- 1. headline einsum (bench.py): trace + three PMC passes ----------------------------------------------------------
cd $ROOT && bash tools/gpu_profile.sh $TAG/einsum > $OUT/einsum_profile.log 2>&1
for d in trace pmc_sq pmc_fetch pmc_write; do cp $OUT/einsum/$d.summary.txt $OUT/einsum_$d.summary.txt 2>/dev/null; done
cp $OUT/einsum/pmc_traffic_einsum.json $OUT/pmc_traffic_einsum.json 2>/dev/null
# ---- 1b. the same headline with HBM-cold operands only (four rotating pairs): trace + PMC of the cold launches alone -------
cd /tmp
COLD="python $ROOT/bench.py --cold-only --steps 500"
rocprofv3 --kernel-trace --stats -d $OUT/cold_trace -o r -- $COLD > $OUT/cold_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/cold_sq -o r -- $COLD > $OUT/cold_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/cold_fetch -o r -- $COLD > $OUT/cold_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/cold_write -o r -- $COLD > $OUT/cold_write.log 2>&1
for p in trace sq fetch write; do summ cold_$p einsum_cold_$p; done
(cd $ROOT && for i in 1 2 3; do python bench.py --cold-only --steps 2000 2>/dev/null | tail -1; done > $OUT/einsum_cold_unprofiled.jsonl)
# ---- 1c. the general MFMA family (round 4): fp64 4096^3 and complex64 2048^3: trace, SQ, FETCH, WRITE ---------------------------
cd /tmp
for W in f64 c64; do
  G="python $ROOT/tools/bench_gen.py $W"
  rocprofv3 --kernel-trace --stats -d $OUT/gen_${W}_trace -o r -- $G > $OUT/gen_${W}_trace.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/gen_${W}_sq -o r -- $G > $OUT/gen_${W}_sq.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/gen_${W}_fetch -o r -- $G > $OUT/gen_${W}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/gen_${W}_write -o r -- $G > $OUT/gen_${W}_write.log 2>&1
  for p in trace sq fetch write; do summ gen_${W}_$p gen_${W}_$p; done
done
(cd $ROOT && python tools/bench_gen.py > $OUT/bench_gen.jsonl 2>/dev/null; CUTENSOR_AMD_GEN=0 python tools/bench_gen.py einsum f64 > $OUT/bench_gen_before.jsonl 2>/dev/null)
# ---- 1d. the mid-size 16-bit kernel (round 4): bf16 2048^3 on gett_h16w4m4_kernel: trace, SQ, FETCH, WRITE -----------------------
cd /tmp
M="python $ROOT/tools/h16_shape_sweep.py --only 2048,2048,2048 --reps 200"
rocprofv3 --kernel-trace --stats -d $OUT/h16m_trace -o r -- $M > $OUT/h16m_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/h16m_sq -o r -- $M > $OUT/h16m_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/h16m_fetch -o r -- $M > $OUT/h16m_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/h16m_write -o r -- $M > $OUT/h16m_write.log 2>&1
for p in trace sq fetch write; do summ h16m_$p h16_mid_2048_$p; done
(cd $ROOT && for L in mk,kn km,kn mk,nk; do python tools/h16_shape_sweep.py --layout $L 2>/dev/null; done > $OUT/h16_shape_sweep.jsonl)
# ---- 1e. the small-problem 16-bit kernel (round 4): bf16 1024^3 on gett_h16w4q_kernel: trace, SQ, FETCH; in-kernel timeline ----------
cd /tmp
S="python $ROOT/tools/h16_shape_sweep.py --only 1024,1024,1024 --reps 400"
rocprofv3 --kernel-trace --stats -d $OUT/h16q_trace -o r -- $S > $OUT/h16q_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/h16q_sq -o r -- $S > $OUT/h16q_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/h16q_fetch -o r -- $S > $OUT/h16q_fetch.log 2>&1
for p in trace sq fetch; do summ h16q_$p h16_small_1024_$p; done
(cd $ROOT && for s in "1024 1024 1024" "1024 1024 4096" "2048 2048 2048"; do python tools/h16_small_timeline.py $s 2>/dev/null; done > $OUT/h16_small_timeline.jsonl)
# ---- 2. bf16 8192^3, default kernel, two layouts: trace, SQ, FETCH, WRITE ---------------------------------------------
cd /tmp
for L in mk,kn km,kn; do
  N=$(echo $L | tr , _)
  H16="python $ROOT/tools/bench_h16.py --layout $L --reps 30"
  rocprofv3 --kernel-trace --stats -d $OUT/h16_${N}_trace -o r -- $H16 > $OUT/h16_${N}_trace.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/h16_${N}_sq -o r -- $H16 > $OUT/h16_${N}_sq.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/h16_${N}_fetch -o r -- $H16 > $OUT/h16_${N}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/h16_${N}_write -o r -- $H16 > $OUT/h16_${N}_write.log 2>&1
  for p in trace sq fetch write; do summ h16_${N}_$p h16_${N}_$p; done
done
# ---- 2b. all four layouts beside the vendor GEMM and the MFMA-only rate of this box, same call (no profiler) ----------------
cd $ROOT
: > $OUT/h16_vs_vendor.jsonl
for L in mk,kn km,kn mk,nk km,nk; do
  python tools/bench_h16.py --layout $L 2>/dev/null | grep workload >> $OUT/h16_vs_vendor.jsonl
  python tools/bench_h16.py --layout $L --zeros 2>/dev/null | grep workload >> $OUT/h16_vs_vendor.jsonl
done
python tools/ubench/vendor_gemm_bf16.py 2>/dev/null | grep vendor >> $OUT/h16_vs_vendor.jsonl
python tools/ubench/vendor_gemm_bf16.py --zeros 2>/dev/null | grep vendor >> $OUT/h16_vs_vendor.jsonl
python - >> $OUT/h16_vs_vendor.jsonl 2>/dev/null <<PY
import ctypes, json, sys
sys.path.insert(0, '.')
from cudalibrarysamples_amd import cutensor as ct
import torch
torch.cuda.init()
out = {}
for name, kind, shape in (('zeros', 0, 1), ('uniform', 1, 1), ('uniform_32x32x16', 1, 0)):   # the default kernel's MFMA shape is 16x16x32
    v = ctypes.c_float(0); ct.lib.ctamdMeasureMfmaCeilingShape(1, kind, shape, ctypes.byref(v)); out[name] = v.value
print(json.dumps({'mfma_only_tflops': out}))
PY
python tools/h16_ksweep.py --zeros 2>/dev/null | tail -1 > $OUT/h16_ksweep.jsonl
python tools/h16_ksweep.py 2>/dev/null | tail -1 >> $OUT/h16_ksweep.jsonl
# ---- 2c. permute / reduce at 2048^3 (BASELINE configs[2]): trace, FETCH_SIZE, WRITE_SIZE + L2/EA request counters ---------
cd /tmp
BW="python $ROOT/tools/bench_bandwidth.py --n 2048 --reps 3"
rocprofv3 --kernel-trace --stats -d $OUT/bw_trace -o r -- $BW > $OUT/bw_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/bw_fetch -o r -- $BW > $OUT/bw_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/bw_write -o r -- $BW > $OUT/bw_write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/bw_tcc -o r -- $BW > $OUT/bw_tcc.log 2>&1
for p in trace fetch write tcc; do summ bw_$p bandwidth_2048_$p; done
cd $ROOT
python tools/bench_bandwidth.py --n 2048 --check > $OUT/bandwidth_2048.jsonl 2>/dev/null
python tools/bench_bandwidth.py --n 1024 --check > $OUT/bandwidth_1024.jsonl 2>/dev/null
samples/bin/einsum --flow --calls 2000 > $OUT/einsum_flow_plan_per_call.jsonl 2>/dev/null
cd /tmp
# ---- 3. the whole default bench line (secondary configs included): kernel trace only ------------------------------------
rocprofv3 --kernel-trace --stats -d $OUT/bench_all_trace -o r -- python $ROOT/bench.py --steps 200 --warmup 20 --no-cpu > $OUT/bench_all_trace.log 2>&1
summ bench_all_trace bench_all_trace
# ---- 4. un-profiled reference lines ------------------------------------------------------------------------------------
cd $ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2>&1
python bench.py > $OUT/bench_default.log 2>&1
find $OUT -name '*.csv' -size +1M -delete
find $OUT -name '*.db' -delete
rm -rf $OUT/einsum/trace $OUT/einsum/pmc_sq $OUT/einsum/pmc_fetch $OUT/einsum/pmc_write
du -sh $OUT
ls $OUT
