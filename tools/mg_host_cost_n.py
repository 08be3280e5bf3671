#!/usr/bin/env python3
"""Host side of ONE cutensorMgContraction call at n LOGICAL devices (all on GPU 0: the machinery — pieces, staging copies, event waits,
stream switches — is what costs host time, and it runs the same on one physical GPU), round-5 review Weak #6: DESIGN.md section 5 models
4096^3 on 8 devices at ~0.17 ms of device time plus "~0.1 ms of single-thread host enqueue" and nobody had measured the second number.
For n in 1, 2, 4, 8 and two layouts — bench.py's free-mode layout (A, C row slabs; B column slabs) and the sample's 2 x 2 block-cyclic
descriptors (contraction_multi_gpu.cu:154-193) — one JSON line: the wall clock of the call alone from an idle device (min / median of 30),
the per-call host time in a run of back-to-back calls on a SMALL problem (the queue never fills, so this is pure enqueue cost), pieces,
local contractions, transfers and local copies per call."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(cm, torch, n, layout, E):
    if layout == "free":
        modes = ["ik", "kj", "ij"]
        block = [dict(i=E // n), dict(j=E // n), dict(i=E // n, j=E // n)]
        dcount = [dict(i=n), dict(j=n), dict(i=n)]
        ncell = [n, n, n]
    else:   # the sample: block E/2 per mode, 2 x 2 device grid per tensor, cells owned cyclically by the handle's devices
        modes = ["ik", "kj", "ij"]
        b = E // 2
        block = [dict(i=b, k=b), dict(k=b, j=b), dict(i=b, j=b)]
        dcount = [dict(i=2, k=2), dict(k=2, j=2), dict(i=2, j=2)]
        ncell = [4, 4, 4]
    devs = [0] * n
    con = cm.Contraction(devs, modes, dict(i=E, j=E, k=E), block, dcount)
    try:
        d = con.describe()
        cells = [[torch.rand(E * E // ncell[k], device="cuda") for _ in range(ncell[k])] for k in range(3)]
        ws = [torch.empty(max(int(con.ws_sizes[g]), 16), dtype=torch.uint8, device="cuda") for g in range(n)]
        streams = [torch.cuda.Stream(device=0) for _ in range(n)]
        ptr = [[t.data_ptr() for t in row] for row in cells]
        wsp, sp = [t.data_ptr() for t in ws], [s.cuda_stream for s in streams]
        call = lambda: cm.check(con.run(1.0, ptr[0], ptr[1], 0.0, ptr[2], ptr[2], wsp, sp))  # noqa: E731
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        idle = []
        for _ in range(30):
            torch.cuda.synchronize()
            time.sleep(0.001)
            t0 = time.perf_counter()
            call()
            idle.append((time.perf_counter() - t0) * 1e6)
        torch.cuda.synchronize()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        nsub = d.get("localContractions", len(d["pieces"]))
        # cross-device waits of the join (step 4).  n LOGICAL devices all name GPU 0, so every handle entry "owns" every cell and each
        # scattering stream is waited for by all n - 1 others; on n physical devices a cell has one owner (cell c -> handle entry c % n here)
        join_logical = sum(len(o) for o in d.get("scatterOwners", []))
        join_physical = sum(len({c % n for c in p.get("scatter", [])} - {p["dev"]}) for p in d["pieces"])
        return {"n_logical_devices": n, "layout": layout, "extent": E, "host_us_idle_min": round(min(idle), 1), "host_us_idle_median": round(statistics.median(idle), 1),
                "host_us_back_to_back": round((t1 - t0) / reps * 1e6, 1), "device_bound": bool((t2 - t1) > 0.2 * (t1 - t0)),
                "pieces": len(d["pieces"]), "local_contractions": nsub, "transfers": len([t for t in d["transfers"] if not t["local"]]),
                "local_copies": len([t for t in d["transfers"] if t["local"]]), "join_waits_logical": join_logical, "join_waits_physical": join_physical,
                "host_us_per_piece": round((t1 - t0) / reps * 1e6 / max(len(d["pieces"]), 1), 2)}
    finally:
        con.close()


def main():
    import torch
    from cudalibrarysamples_amd import cutensormg as cm
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    for layout in ("free", "sample"):
        for n in (1, 2, 4, 8):
            try:
                print(json.dumps(measure(cm, torch, n, layout, E)), flush=True)
            except Exception as ex:  # noqa: BLE001
                print(json.dumps({"n_logical_devices": n, "layout": layout, "error": "%s: %s" % (type(ex).__name__, ex)}), flush=True)


if __name__ == "__main__":
    main()
