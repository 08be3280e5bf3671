#!/usr/bin/env python3
"""Per-workgroup timeline of gett_h16w4q_kernel (the 64 x 64 tile) on a small bf16 problem (TIMED instantiation,
CUTENSOR_AMD_H16_TIMED=1, layout mk,kn): where do the ~7 us go that a launch costs beyond its K-tiles?
usage: python tools/h16_small_timeline.py [M N K]"""
import json
import os
import sys

os.environ["CUTENSOR_AMD_H16_TIMED"] = "1"
os.environ.setdefault("CUTENSOR_AMD_H16_WAVES", "4q")
os.environ["CUTENSOR_AMD_H16_SPLITK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cudalibrarysamples_amd import cutensor as ct, ops

M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else (1024, 1024, 1024)
A = (torch.rand((K, M), device="cuda") * 2 - 1).to(torch.bfloat16)      # "mk": m fastest
B = (torch.rand((N, K), device="cuda") * 2 - 1).to(torch.bfloat16)      # "kn"
D = torch.empty((N, M), device="cuda", dtype=torch.bfloat16)
h = ops.Handle()
plan = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF)
nwg = ((M + 63) // 64) * ((N + 63) // 64)
tbuf = torch.zeros(64 + 8 * nwg, dtype=torch.int64, device="cuda")
for _ in range(200):
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
torch.cuda.synchronize()
ct.lib.ctamdSetTimingBuffer(h.h, tbuf.data_ptr())
for _ in range(3):      # the last of three back-to-back launches is the one recorded (a warm device, a predecessor in flight)
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
torch.cuda.synchronize()
ct.lib.ctamdSetTimingBuffer(h.h, None)
t = tbuf.cpu().numpy()[64:].reshape(nwg, 8).astype(np.float64)
seg = {"arguments fetched + tile located": t[:, 7] - t[:, 0], "setup(entry->first piece)": t[:, 1] - t[:, 0], "first tile lands": t[:, 2] - t[:, 1], "main loop": t[:, 3] - t[:, 2],
       "epilogue (stores issued)": t[:, 4] - t[:, 3]}
w0 = t[:, 5].min()
start, end = (t[:, 5] - w0) / 100.0, (t[:, 6] - w0) / 100.0          # wall clock: 100 MHz -> us
clk = (t[:, 4] - t[:, 0]) / ((end - start) * 1e3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
e1.record()
torch.cuda.synchronize()
out = {"shape": [M, N, K], "kname": plan.describe()["kname"], "workgroups": nwg, "k_tiles": K // 64,
       "cycles_mean": {k: float(v.mean()) for k, v in seg.items()}, "cycles_max": {k: float(v.max()) for k, v in seg.items()},
       "clock_ghz_mean": float(clk.mean()), "wg_start_us": [float(start.min()), float(np.median(start)), float(start.max())],
       "wg_end_us": [float(end.min()), float(np.median(end)), float(end.max())], "wg_dur_us_mean": float((end - start).mean()),
       "us_per_call_200_back_to_back": e0.elapsed_time(e1) * 1e3 / 200}
print(json.dumps(out))
