#!/bin/bash
# Round 5, GPU call 9 (RESEARCH build in lib/): wait accounting of the headline kernel, warm and HBM-cold (review item 6).
set -u
OUT=gpurun_out/r05i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/headline_cold_counters.py > $OUT/headline_cold_counters.json 2> $OUT/headline_cold_counters.err; echo "rc $?"; cat $OUT/headline_cold_counters.json; tail -3 $OUT/headline_cold_counters.err
