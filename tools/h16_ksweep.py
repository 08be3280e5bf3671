#!/usr/bin/env python3
"""Fixed vs per-K-tile cost of a 16-bit kernel variant: M = N = 8192, K swept (zero-filled or U(-1,1) operands).
usage: [CUTENSOR_AMD_H16_WAVES=4|s] [CUTENSOR_AMD_H16_ABL=n] python tools/h16_ksweep.py [--zeros]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops

zeros = "--zeros" in sys.argv
h = ops.Handle()
n = 8192
rows = []
for K in (64, 512, 2048, 8192):
    A = torch.zeros((n, K), device="cuda", dtype=torch.bfloat16) if zeros else (torch.rand((n, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    B = torch.zeros((n, K), device="cuda", dtype=torch.bfloat16) if zeros else (torch.rand((n, K), device="cuda") * 2 - 1).to(torch.bfloat16)
    D = torch.empty((n, n), device="cuda", dtype=torch.bfloat16)
    p = ops.contraction_plan(h, [K, n], "km", [K, n], "kn", [n, n], "mn", dtype=ct.R_16BF)
    for _ in range(30):
        p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    rows.append((K, ms))
    kn = p.describe()["kname"]
per_tile_us = (rows[3][1] - rows[2][1]) * 1e3 / ((8192 - 2048) / 64) / 4      # 4 tile rounds per CU
fixed_us = rows[2][1] * 1e3 / 4 - per_tile_us * 2048 / 64
print(json.dumps({"kernel": kn, "zeros": zeros, "abl": os.environ.get("CUTENSOR_AMD_H16_ABL", "0"), "ms": rows,
                  "per_64k_tile_us": per_tile_us, "fixed_us_per_workgroup": fixed_us,
                  "per_tile_cycles_at_2p4GHz": per_tile_us * 2400}))
