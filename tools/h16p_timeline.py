#!/usr/bin/env python3
"""Per-workgroup timeline of the persistent 16-bit kernel (gett_h16w4p_kernel, TIMED instantiation: CUTENSOR_AMD_H16_TIMED=1, bf16 mk,kn):
shader cycles from entry to the first tile's data (setup + first LDS-DMA burst), in its main loop, in its epilogue, and per tile over
the workgroup's whole walk.  usage: [CUTENSOR_AMD_H16P_EP=0|1|2] python tools/h16p_timeline.py M N K [--zeros]"""
import json
import os
import sys

os.environ["CUTENSOR_AMD_H16_TIMED"] = "1"
os.environ["CUTENSOR_AMD_H16_WAVES"] = "4p"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cudalibrarysamples_amd import cutensor as ct, ops

args = [a for a in sys.argv[1:] if not a.startswith("--")]
M, N, K = (int(x) for x in args[:3]) if len(args) >= 3 else (8192, 8192, 8192)
zeros = "--zeros" in sys.argv
mk = lambda r, c: torch.zeros((r, c), device="cuda", dtype=torch.bfloat16) if zeros else (torch.rand((r, c), device="cuda") * 2 - 1).to(torch.bfloat16)  # noqa: E731
A, B = mk(K, M), mk(N, K)                  # "mk" (m fastest) and "kn"
D = torch.empty((N, M), device="cuda", dtype=torch.bfloat16)
h = ops.Handle()
plan = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF, workspace_limit=0)
d = plan.describe()
assert d["kname"] == "gett_h16w4p_kernel", d
nwg = min(256, (M // 256) * (N // 256))
tbuf = torch.zeros(64 + 8 * 1024, dtype=torch.int64, device="cuda")
fn = lambda: plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())  # noqa: E731
for _ in range(40):
    fn()
torch.cuda.synchronize()
ct.lib.ctamdSetTimingBuffer(h.h, tbuf.data_ptr())
fn()
torch.cuda.synchronize()
ct.lib.ctamdSetTimingBuffer(h.h, None)
ref = (B[:256].float() @ A.float())
err = float((D[:256].float() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) if not zeros else float(D.float().abs().max())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
t = tbuf.cpu().numpy()[64:64 + 8 * nwg].reshape(nwg, 8).astype(np.float64)
tiles = t[:, 5]
out = {"M": M, "N": N, "K": K, "zeros": zeros, "ep": os.environ.get("CUTENSOR_AMD_H16P_EP", "2"), "rel_err_256_rows": err, "ms": ms,
       "tflops_20_calls": 2.0 * M * N * K / (ms * 1e-3) / 1e12, "workgroups": nwg, "tiles_per_workgroup": [float(tiles.min()), float(tiles.max())],
       "cycles_mean": {"entry_to_first_tile_landed": float((t[:, 1] - t[:, 0]).mean()), "main_loop_tile0": float((t[:, 2] - t[:, 1]).mean()),
                       "epilogue_tile0_incl_next_setup": float((t[:, 3] - t[:, 2]).mean()), "whole_walk": float((t[:, 4] - t[:, 0]).mean()),
                       "per_tile_after_the_first": float(((t[:, 4] - t[:, 3]) / np.maximum(tiles - 1, 1)).mean()) if tiles.max() > 1 else None},
       "k_tiles": K // 64}
print(json.dumps(out))
