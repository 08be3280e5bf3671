#!/usr/bin/env python3
"""Fixed vs per-K-tile cost of the fp32 GETT kernel the planner picks for C[m,n] = A[m,k] B[k,n] (the cuTENSORMg sample's local
contraction): M = N fixed, K swept, no split-K (workspace 0).  time = rounds x (fixed + K-tiles x slope).
usage: python tools/f32_ksweep.py [--n 4096] [--layout ik,kj] [--algo r]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--layout", default="mk,kn")
ap.add_argument("--algo", type=int, default=None)
args = ap.parse_args()
mA, mB = args.layout.split(",")
h = ops.Handle()
n = args.n
rows = []
desc = None
for K in (32, 256, 1024, 4096):
    ext = {"m": n, "n": n, "k": K}
    A = torch.rand([ext[c] for c in reversed(mA)], device="cuda") * 2 - 1
    B = torch.rand([ext[c] for c in reversed(mB)], device="cuda") * 2 - 1
    D = torch.empty((n, n), device="cuda")
    kw = {"workspace_limit": 0}
    if args.algo is not None:
        kw["algo"] = args.algo
    p = ops.contraction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [n, n], "mn", **kw)
    desc = p.describe()
    for _ in range(20):
        p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    rows.append((K, e0.elapsed_time(e1) / 30, desc["kname"], desc["bm"], desc["bn"], desc["bk"], desc["blocks"]))
bm, bn, bk, blocks = rows[-1][3], rows[-1][4], rows[-1][5], rows[-1][6]
rounds = max(1.0, blocks / 256.0)
slope = (rows[3][1] - rows[2][1]) * 1e3 / ((4096 - 1024) / bk) / rounds
fixed = rows[2][1] * 1e3 / rounds - slope * 1024 / bk
ideal = 2.0 * bm * bn * bk / (157.2864e12 / 256) * 1e6
print(json.dumps({"n": n, "layout": args.layout, "kernel": rows[-1][2], "tile": [bm, bn, bk], "workgroups": blocks, "rounds": rounds,
                  "ms": [(r[0], round(r[1], 5)) for r in rows], "per_k_tile_us": slope, "ideal_per_k_tile_us": ideal,
                  "fixed_us_per_workgroup": fixed, "tflops_at_4096": 2.0 * n * n * 4096 / rows[3][1] / 1e9}))
