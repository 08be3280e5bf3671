#!/usr/bin/env python3
"""Per-launch timeline from a rocprofv3 rocpd database: duration of the engine's kernels and the gap to the next launch,
averaged over windows of launches — shows clock ramp-up / overlap effects over a long run.  usage: trace_timeline.py <db> [window]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    win = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = list(c.execute("select name, start, end from kernels order by start"))
    eng = [(n, s, e) for (n, s, e) in rows if "ctamd" in n]
    print("# %d engine launches; columns of `kernels`: %s" % (len(eng), ",".join(cols)))
    gett = [i for i, r in enumerate(eng) if "gett" in r[0]]
    for w0 in range(0, len(gett) - 1, win):
        idx = gett[w0:w0 + win]
        if len(idx) < 2:
            break
        dur_g = sum(eng[i][2] - eng[i][1] for i in idx) / len(idx) / 1e3
        folds = [i + 1 for i in idx if i + 1 < len(eng) and "reduce" in eng[i + 1][0]]
        dur_f = sum(eng[i][2] - eng[i][1] for i in folds) / max(len(folds), 1) / 1e3
        gap_gf = sum(eng[i][1] - eng[i - 1][2] for i in folds) / max(len(folds), 1) / 1e3
        nxt = [i + 1 for i in folds if i + 1 < len(eng)]
        gap_fg = sum(eng[i][1] - eng[i - 1][2] for i in nxt) / max(len(nxt), 1) / 1e3
        period = (eng[idx[-1]][1] - eng[idx[0]][1]) / (len(idx) - 1) / 1e3
        print("launch %5d..%5d  t=%8.2f ms  gett %.2f us  fold %.2f us  gap gett->fold %.2f  fold->gett %.2f  period %.2f us" % (
            w0, w0 + len(idx) - 1, (eng[idx[0]][1] - eng[0][1]) / 1e6, dur_g, dur_f, gap_gf, gap_fg, period))


if __name__ == "__main__":
    main()
