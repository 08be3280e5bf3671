#!/bin/bash
# Round 5, GPU call 10 (RESEARCH build in lib/): the headline under other orders of its contracted digits (CUTENSOR_AMD_KORDER):
# which K-tiles concurrent workgroups read decides how the operand stream meets DRAM.
set -u
OUT=gpurun_out/r05k; mkdir -p $OUT
export TMPDIR=/tmp
for ko in "" "d,b:16,c,b" "d,b:16,b,c" "d,b:32,c,b" "d,c:16,b,c" "d:32,b:32,c,b"; do
  if [ -z "$ko" ]; then timeout 300 python tools/headline_cold_counters.py >> $OUT/korder.jsonl 2>> $OUT/korder.err
  else CUTENSOR_AMD_KORDER="$ko" timeout 300 python tools/headline_cold_counters.py >> $OUT/korder.jsonl 2>> $OUT/korder.err; fi
done
python - <<'PY'
import json
for l in open("gpurun_out/r05k/korder.jsonl"):
    d = json.loads(l)
    if "error" in d: print(d); continue
    print(d["korder"], d["Kdigits"], "err %.1e" % d["max_rel_err"], {k: (d[k]["warm_us"], d[k]["cold_us"]) for k in ("default", "nt_twin", "abl2_no_mfma") if k in d})
PY
tail -3 $OUT/korder.err
