#!/usr/bin/env python3
"""In-kernel phase timing of the headline einsum: per-workgroup timestamps written by the GETT kernel
(ctamdSetTimingBuffer) -> where do the cycles of one launch go?"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", type=int, default=-1)
    ap.add_argument("--b", type=int, default=64)
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--dump", type=str, default="", help="save the raw per-workgroup timing records (.npy)")
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    ext = dict(a=96, b=args.b, c=64, d=64, e=96)
    mA, mB, mC = "dcba", "ebcd", "ea"
    eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
    h = ops.Handle()
    A = torch.rand(int(np.prod(eA)), device="cuda")
    B = torch.rand(int(np.prod(eB)), device="cuda")
    C = torch.zeros(int(np.prod(eC)), device="cuda")
    kw = dict(workspace_limit=1 << 30)
    if args.algo >= 0:
        kw["algo"] = args.algo
    p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, **kw)
    d = p.describe()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    tbuf = torch.zeros(d["blocks"] * 16, dtype=torch.int64, device="cuda")
    for _ in range(5 if args.cold else 1500):   # steady clock state by default; --cold: 5 warm-up calls
        p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), 1 << 30, 0)
    torch.cuda.synchronize()
    ct.lib.ctamdSetTimingBuffer(h.h, tbuf.data_ptr())
    p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), 1 << 30, 0)
    torch.cuda.synchronize()
    ct.lib.ctamdSetTimingBuffer(h.h, None)
    t = tbuf.cpu().numpy().reshape(-1, 16).astype(np.float64)
    if args.dump:
        np.save(args.dump, tbuf.cpu().numpy().reshape(-1, 16))
    cyc = t[:, :5]
    phases = np.diff(cyc, axis=1)          # prologue, steady, drain, epilogue (shader cycles)
    wall0, wall1 = t[:, 5], t[:, 6]        # 100 MHz wall clock
    out = {
        "plan": d,
        "cycles_mean": dict(zip(["prologue", "steady", "drain", "epilogue"], [float(x) for x in phases.mean(axis=0)])),
        "cycles_max": dict(zip(["prologue", "steady", "drain", "epilogue"], [float(x) for x in phases.max(axis=0)])),
        "total_cycles_mean": float((cyc[:, 4] - cyc[:, 0]).mean()),
        "wait_cycles_mean": {"multiplier_barrier": float(t[:, 8].mean()), "loader_vmcnt": float(t[:, 9].mean()),
                             "loader_barrier": float(t[:, 10].mean())} if (t[:, 8] > 0).any() else None,
        "setup_cycles_mean": float((t[:, 7] - t[:, 0]).mean()) if (t[:, 7] > 0).all() else None,   # entry -> first LDS-DMA issue (streaming kernels)
        "wall_us_first_start_to_last_end": float((wall1.max() - wall0.min()) / 100.0),
        "wall_us_start_skew": float((wall0.max() - wall0.min()) / 100.0),
        "wall_us_end_skew": float((wall1.max() - wall1.min()) / 100.0),
        "wall_us_per_wg_mean": float((wall1 - wall0).mean() / 100.0),
        "clock_ghz_est": float(((cyc[:, 4] - cyc[:, 0]) / ((wall1 - wall0) * 10.0)).mean()),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
