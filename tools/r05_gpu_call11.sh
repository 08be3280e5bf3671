#!/bin/bash
# Round 5, GPU call 11: the whole GPU suite on the production build, smoke, every fuzzer once.
set -u
OUT=gpurun_out/r05l; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/summary.txt; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/summary.txt
f() { name=$1; shift; timeout 400 python "$@" > $OUT/fuzz_$name.log 2>&1; echo "fuzz $name rc $? : $(tail -1 $OUT/fuzz_$name.log | cut -c1-260)" | tee -a $OUT/summary.txt; }
f contraction tools/fuzz_contraction.py --cases 300 --seed 11
f contraction_all_types tools/fuzz_contraction.py --all-types --cases 250 --seed 12
f contraction_strided tools/fuzz_contraction.py --all-types --strided --cases 200 --seed 13
f contraction_many_modes tools/fuzz_contraction.py --many-modes --cases 120 --seed 14
f contraction_aligned tools/fuzz_contraction.py --aligned --cases 200 --seed 15
f contraction_ragged_k tools/fuzz_contraction.py --ragged-k --strided --cases 200 --seed 16
f elementwise tools/fuzz_elementwise.py --cases 500 --seed 17
f elementwise_wide tools/fuzz_elementwise.py --wide --cases 150 --seed 18
f einsum tools/fuzz_einsum.py --cases 200 --seed 19
f stream tools/fuzz_stream.py --cases 80 --seed 20
f mg tools/fuzz_mg.py --cases 40 --seed 21
f mg_ragged tools/fuzz_mg.py --ragged --cases 30 --seed 22
f mp tools/fuzz_mp.py --cases 100 --seed 23
