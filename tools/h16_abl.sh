#!/bin/bash
# Ablation sweep of the four-wave 16-bit kernel (measurement only).  usage: tools/h16_abl.sh "<abl list>" [layout] [extra bench args]
export CUTENSOR_AMD_H16_WAVES=${WAVES:-4}
for a in $1; do
    CUTENSOR_AMD_H16_ABL=$a timeout 120 python tools/bench_h16.py --layout ${2:-km,kn} $3 2>&1 | grep workload > /tmp/abl.json
    python - "$a" <<'PY'
import json, sys
d = json.loads(open("/tmp/abl.json").read())
print("abl %s: ms %.4f TF %.0f (kernel %s)" % (sys.argv[1], d["ms_per_call"], d["tflops"], d["plan"]["kernel"]))
PY
done
