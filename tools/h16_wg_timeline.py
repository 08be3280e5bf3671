#!/usr/bin/env python3
"""Per-workgroup timeline of the default 16-bit GETT kernel on 8192^3 (TIMED instantiation, CUTENSOR_AMD_H16_TIMED=1, layout
mk,kn): shader cycles spent before the main loop / in it / in the epilogue, and the wall-clock start and end of every
workgroup -> where does a launch's time go beyond 4 rounds x 128 K-tiles?  usage: python tools/h16_wg_timeline.py [--zeros]"""
import json
import os
import sys

os.environ["CUTENSOR_AMD_H16_TIMED"] = "1"          # (TIMED / XST instantiations: research builds only, make RESEARCH=1)
os.environ.setdefault("CUTENSOR_AMD_H16_WAVES", "4x")   # the one-tile kernel (the planner takes its persistent form for 8192^3 since round 5)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cudalibrarysamples_amd import cutensor as ct, ops

zeros = "--zeros" in sys.argv
n = 8192
A = torch.zeros((n, n), device="cuda", dtype=torch.bfloat16) if zeros else (torch.rand((n, n), device="cuda") * 2 - 1).to(torch.bfloat16)
B = torch.zeros((n, n), device="cuda", dtype=torch.bfloat16) if zeros else (torch.rand((n, n), device="cuda") * 2 - 1).to(torch.bfloat16)
D = torch.empty((n, n), device="cuda", dtype=torch.bfloat16)
h = ops.Handle()
plan = ops.contraction_plan(h, [n, n], "mk", [n, n], "kn", [n, n], "mn", dtype=ct.R_16BF)
nwg = 1024
tbuf = torch.zeros(64 + 8 * nwg, dtype=torch.int64, device="cuda")
for _ in range(60):
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
torch.cuda.synchronize()
ct.lib.ctamdSetTimingBuffer(h.h, tbuf.data_ptr())
plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
torch.cuda.synchronize()
ct.lib.ctamdSetTimingBuffer(h.h, None)
# the measurement instantiations (CUTENSOR_AMD_H16_XST) must still compute the product: 256 rows against torch, and the rate of 20 calls
# (modes "mk" / "kn" / "mn" name the FASTEST mode first: the torch tensors are [k][m], [n][k] and [n][m])
ref = (B[:256].float() @ A.float())
err = float((D[:256].float() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) if not zeros else float(D.float().abs().max())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
e1.record()
torch.cuda.synchronize()
tflops = 2.0 * n ** 3 * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e12
t = tbuf.cpu().numpy()[64:].reshape(nwg, 8).astype(np.float64)
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
w0 = t[:, 4].min()
start, end = (t[:, 4] - w0) / 100.0, (t[:, 5] - w0) / 100.0         # wall clock: 100 MHz -> us
dur = end - start
clk = (t[:, 3] - t[:, 0]) / (dur * 1e3)                               # GHz
order = np.argsort(start)
rounds = [order[i * 256:(i + 1) * 256] for i in range(4)]
extra = {}
if os.environ.get("CUTENSOR_AMD_H16_XST") == "7":      # slot 1 = cycles in HEpilogue::init, slot 7 = cycles in the epilogue's LDS-write phases
    extra = {"epilogue_init_cycles": float(t[:, 1].mean()), "epilogue_lds_write_phase_cycles": float(t[:, 7].mean())}
    pro = pro * 0
out = {"zeros": zeros, **extra, "xst": os.environ.get("CUTENSOR_AMD_H16_XST", "0"), "rel_err_256_rows": err, "tflops_20_calls": tflops,
       "cycles_mean": {"prologue": pro.mean(), "main_loop": loop.mean(), "epilogue": epi.mean()},
       "cycles_per_k_tile": loop.mean() / 128, "clock_ghz_mean": clk.mean(), "kernel_span_us": float(end.max()),
       "rounds": [{"start_us": [float(start[r].min()), float(start[r].max())], "end_us": [float(end[r].min()), float(end[r].max())],
                   "dur_us_mean": float(dur[r].mean())} for r in rounds],
       "per_xcd_dur_us": [float(dur[t[:, 6] == x].mean()) for x in range(8)]}
print(json.dumps(out))
