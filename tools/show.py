#!/usr/bin/env python3
"""Print tune_gett.py / phase_timing.py JSON lines compactly (helper for reading gpurun_out/)."""
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        l = l.strip()
        if not l.startswith("{"): continue
        d = json.loads(l)
        if "rank" in d:
            print("k%-3d abl%d S%-4d us %.1f kern %.1f min %.1f" % (d["kernel"], d["abl"], d["splitK"], d["us"], d["kernel_us"], d["kernel_min_us"]), d["tile"], "pf", d["pf"])
        elif "cycles_mean" in d:
            p = d["plan"]
            print("k%-3d abl%d" % (p["kernel"], p["abl"]), {k: round(v) for k, v in d["cycles_mean"].items()}, "tot", round(d["total_cycles_mean"]),
                  {k: round(v) for k, v in (d["wait_cycles_mean"] or {}).items()}, "setup", round(d["setup_cycles_mean"] or 0), "wall", d["wall_us_first_start_to_last_end"],
                  "skew", d["wall_us_start_skew"], d["wall_us_end_skew"], "clk", round(d["clock_ghz_est"], 3))
