#!/bin/bash
# Round 5, GPU call 15: the whole GPU suite, smoke and the driver-style bench line on the final production build.
set -u
OUT=gpurun_out/r05x; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -n "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.log 2> $OUT/bench_steps20.err; echo "bench rc $? lines $(wc -l < $OUT/bench_steps20.log)"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05x/bench_steps20.log").read().strip().splitlines()[-1])
print("value %.1f" % (d["value"] / 1e3), "frac", round(d["roofline"]["frac"], 4))
for s in d["secondary"][:2]:
    print(s["workload"][:60], round(s["value"] / 1e3, 1), s.get("kernel"), round(s["roofline"]["frac"], 4))
PY
