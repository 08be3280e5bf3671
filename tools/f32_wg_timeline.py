#!/usr/bin/env python3
"""Per-workgroup phases of the fp32 streaming GETT kernel on a GEMM-like contraction that stores D itself (no split-K): in-kernel
stamps (ctamdSetTimingBuffer) -> prologue / steady K-tiles / drain / epilogue in shader cycles, workgroup start and end on the wall
clock.  The cuTENSORMg sample's 4096^3 on one device is this shape.  usage: python tools/f32_wg_timeline.py [M N K]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops

M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else (4096, 4096, 4096)
h = ops.Handle()
A = torch.rand(M * K, device="cuda")
B = torch.rand(K * N, device="cuda")
D = torch.zeros(M * N, device="cuda")
algo = os.environ.get("F32_TIMELINE_ALGO")   # rank among the planner's candidates (tools/tune_gett.py lists them); unset: its choice
p = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", workspace_limit=1 << 30, **({"algo": int(algo)} if algo else {}))
d = p.describe()
ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace, 0)   # noqa: E731
for _ in range(50):
    fn()
torch.cuda.synchronize()
tbuf = torch.zeros(d["blocks"] * 16, dtype=torch.int64, device="cuda")
ct.lib.ctamdSetTimingBuffer(h.h, tbuf.data_ptr())
fn()
torch.cuda.synchronize()
ct.lib.ctamdSetTimingBuffer(h.h, None)
t = tbuf.cpu().numpy().reshape(-1, 16).astype(np.float64)
t = t[t[:, 0] > 0]
ph = np.diff(t[:, :5], axis=1)
w0 = t[:, 5].min()
start, end = (t[:, 5] - w0) / 100.0, (t[:, 6] - w0) / 100.0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(json.dumps({"shape": [M, N, K], "algo": algo, "pf": d["pf"], "kname": d["kname"], "tile": [d["bm"], d["bn"], d["bk"]], "blocks": d["blocks"], "splitK": d["splitK"],
                  "cycles_mean": dict(zip(["prologue", "steady", "drain", "epilogue"], [float(x) for x in ph.mean(axis=0)])),
                  "setup_cycles_mean": float((t[:, 7] - t[:, 0]).mean()) if (t[:, 7] > 0).all() else None,
                  "total_cycles_mean": float((t[:, 4] - t[:, 0]).mean()), "wg_dur_us_mean": float((end - start).mean()),
                  "clock_ghz": float(((t[:, 4] - t[:, 0]) / ((end - start) * 1e3)).mean()), "kernel_span_us": float(end.max()),
                  "ms_per_call": ms, "tflops": 2.0 * M * N * K / (ms * 1e-3) / 1e12}))
