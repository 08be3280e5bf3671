#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel duration stats and, if present, the mean
PMC counter values per dispatch for the engine's kernels.  Usage: rocprof_summary.py <db> [<db> ...]"""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name[:110]


def traffic_json(out, dbs):
    """--traffic-json OUT db...: HBM bytes per launch of the dominant engine kernel from the FETCH_SIZE /
    WRITE_SIZE passes, corrected as MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KiB
    and on gfx950 FETCH_SIZE tallies a wide coalesced read stream at half its bytes -> x2."""
    import json
    vals = {}
    for db in dbs:
        c = sqlite3.connect(db)
        try:
            rows = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                  "where kernel_name like '%ctamd%gett%' and counter_name in ('FETCH_SIZE','WRITE_SIZE') "
                                  "group by kernel_name, counter_name"))
        except sqlite3.Error:
            rows = []
        for kname, cname, val, n in rows:
            vals[cname] = {"kernel": short(kname), "mean_KiB_per_launch": val, "launches": n}
    if "FETCH_SIZE" in vals:
        fetch = 2.0 * 1024.0 * vals["FETCH_SIZE"]["mean_KiB_per_launch"]
        write = 1024.0 * vals.get("WRITE_SIZE", {}).get("mean_KiB_per_launch", 0.0)
        vals["hbm_bytes_per_launch"] = fetch + write
        vals["read_bytes_per_launch"] = fetch
        vals["write_bytes_per_launch"] = write
        vals["correction"] = "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read half count) + WRITE_SIZE KiB x 1024"
    with open(out, "w") as f:
        json.dump(vals, f, indent=1)
    print(json.dumps(vals))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--traffic-json":
        return traffic_json(sys.argv[2], sys.argv[3:])
    for db in sys.argv[1:]:
        c = sqlite3.connect(db)
        print("# %s" % db)
        rows = list(c.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration) from kernels "
                              "group by name order by sum(duration) desc"))
        tot = sum(r[5] for r in rows) or 1
        print("%-112s %6s %10s %10s %10s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
        for name, n, avg, mn, mx, sm in rows[:8]:
            print("%-112s %6d %10.2f %10.2f %10.2f %6.1f" % (short(name), n, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * sm / tot))
        try:
            pm = list(c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                "where kernel_name like '%ctamd%' group by kernel_name, counter_name"))
        except sqlite3.Error:
            pm = []
        for kname, cname, val, n in pm:
            print("PMC %-70s %-28s mean_per_dispatch %.6g (n=%d)" % (short(kname)[:70], cname, val, n))


if __name__ == "__main__":
    main()
