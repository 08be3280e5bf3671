#!/bin/bash
# rocprofv3 --kernel-trace of one command; prints per-kernel call counts and average / minimum durations.  usage: tools/trace_one.sh <cmd...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; D=/tmp/trace_one.$$; mkdir -p $D; cd /tmp
rocprofv3 --kernel-trace --stats -d $D -o r -- "$@" > $D/log.txt 2>&1
grep "^{" $D/log.txt
python - $D <<'PY'
import sqlite3, sys, glob, os
for db in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    c = sqlite3.connect(db)
    for name, n, avg, mn in c.execute("select name, count(*), avg(duration), min(duration) from kernels group by name order by sum(duration) desc limit 8"):
        print("trace | %-100s | calls %d | avg_us %.2f | min_us %.2f" % (name[:100], n, avg / 1e3, mn / 1e3))
PY
rm -rf $D
