#!/usr/bin/env python3
"""Sweep every ranked (kernel, split-K) candidate of a contraction on the GPU and print one JSON line
per candidate (time per call from torch/HIP events on the launch stream).  Used to calibrate the
planner's cost model; results are copied into profiles/."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PROBLEMS = {
    "einsum": (dict(a=96, b=64, c=64, d=64, e=96), "dcba", "ebcd", "ea"),
    "einsum48": (dict(a=96, b=48, c=64, d=64, e=96), "dcba", "ebcd", "ea"),
    "contraction": (dict(m=96, n=96, u=96, v=64, h=64, k=64), "mhkn", "ukvh", "munv"),
    "gemm4096": (dict(i=4096, j=4096, k=4096), "ik", "kj", "ij"),
    # round 6: short contracted ranges, batched (tools/bench_einsum_shapes.py; modes listed first-fastest)
    "scores128": (dict(b=8, h=8, q=2048, k=2048, d=128), "dqhb", "dkhb", "kqhb"),      # bhqd,bhkd->bhqk
    "scores64": (dict(b=4, h=16, q=1024, k=1024, d=64), "dqhb", "dkhb", "kqhb"),
    "flat128": (dict(i=16384, j=16384, k=128), "ki", "jk", "ji"),                     # ik,kj->ij
    "batch256": (dict(b=512, i=256, j=256, k=256), "jib", "kjb", "kib"),              # bij,bjk->bik
    "skinny": (dict(i=8192, j=128, k=8192), "ki", "jk", "ji"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", default="einsum")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--max", type=int, default=64)
    ap.add_argument("--splits", type=str, default="", help="comma list of split-K values to keep (empty = all)")
    ap.add_argument("--tiles", type=str, default="", help="comma list of bm values to keep (empty = all)")
    ap.add_argument("--kernels", type=str, default="", help="comma list of kernel table indices to keep (empty = all)")
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    ext, mA, mB, mC = PROBLEMS[args.problem]
    eA, eB, eC = [ext[c] for c in mA], [ext[c] for c in mB], [ext[c] for c in mC]
    flop = 2.0 * np.prod([float(v) for v in ext.values()])
    h = ops.Handle()
    A = torch.rand(int(np.prod(eA)), device="cuda")
    B = torch.rand(int(np.prod(eB)), device="cuda")
    C = torch.zeros(int(np.prod(eC)), device="cuda")
    p0 = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 30)
    n = ct.lib.ctamdCountCandidates(h.h, p0.op, 1 << 30)
    print(json.dumps({"problem": args.problem, "candidates": n, "default": p0.describe()}), flush=True)
    ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    keep_s = {int(x) for x in args.splits.split(",") if x}
    keep_t = {int(x) for x in args.tiles.split(",") if x}
    keep_k = {int(x) for x in args.kernels.split(",") if x}
    done = 0
    for r in range(n):
        if done >= args.max:
            break
        p = ops.contraction_plan(h, eA, mA, eB, mB, eC, mC, workspace_limit=1 << 30, algo=r)
        d0 = p.describe()
        if (keep_s and d0["splitK"] not in keep_s) or (keep_t and d0["bm"] not in keep_t) or (keep_k and d0["kernel"] not in keep_k):
            p.destroy()
            continue
        done += 1
        for _ in range(3):
            p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), 1 << 30, 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream().cuda_stream
        e0.record()
        for _ in range(args.reps):
            p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), 1 << 30, stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        # GETT kernel alone (HIP events recorded by the library around the kernel launch)
        import ctypes
        ct.lib.ctamdProfileBegin(h.h)
        for _ in range(args.reps):
            p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), 1 << 30, stream)
        torch.cuda.synchronize()
        mean_ms, min_ms = ctypes.c_float(0), ctypes.c_float(0)
        ct.lib.ctamdProfileEnd(h.h, ctypes.byref(mean_ms), ctypes.byref(min_ms))
        d = p.describe()
        print(json.dumps({"rank": r, "us": us, "kernel_us": mean_ms.value * 1e3, "kernel_min_us": min_ms.value * 1e3, "tflops": flop / us / 1e6, "kernel": d["kernel"],
                          "tile": [d["bm"], d["bn"], d["bk"]], "waves": [d["wm"], d["wn"], d["wk"]],
                          "pf": d["pf"], "abl": d["abl"], "Kdigits": d.get("Kdigits"), "splitK": d["splitK"], "blocks": d["blocks"], "model_us": d["model_us"]}), flush=True)
        p.destroy()


if __name__ == "__main__":
    main()
