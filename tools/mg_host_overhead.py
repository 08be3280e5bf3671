import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
for n in (1, 2, 4, 8):
    for E in (1024, 4096):
        r = bench.mg_measure(n, E, 50, 5, check=True, virtual=True)
        print(json.dumps({"n": n, "E": r["extent"], "ms_per_step": round(r["ms_per_step"], 4), "gflops": round(r["gflops"]), "pieces": r["pieces"], "transport": r["transport"], "err": r["max_rel_err_sampled"]}))
