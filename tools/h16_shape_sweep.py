#!/usr/bin/env python3
"""bf16 C[m,n] = A[m,k] B[k,n] over a list of shapes with whatever 16-bit kernel CUTENSOR_AMD_H16_WAVES selects (default: the
planner's choice): one JSON line per shape (ms, TFLOP/s, kernel, split-K).  Used to check that a new default kernel does not lose on
shapes other than the 8192^3 headline.  usage: [CUTENSOR_AMD_H16_WAVES=8] tools/h16_shape_sweep.py [--layout mk,kn]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(8192, 8192, 8192), (4096, 4096, 4096), (2048, 2048, 2048), (1024, 1024, 1024), (2048, 2048, 16384), (8192, 8192, 512),
          (512, 512, 65536), (256, 256, 16384), (4096, 1024, 4096), (1000, 1000, 1024)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="mk,kn")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--only", default="", help="M,N,K[;M,N,K...]: these shapes only")
    ap.add_argument("--beta", type=float, default=0.0, help="beta of D = A B + beta C (C = D: in place); the persistent kernel streams tiles only for beta = 0")
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    mA, mB = args.layout.split(",")
    h = ops.Handle()
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    shapes = [tuple(int(x) for x in sh.split(",")) for sh in args.only.split(";")] if args.only else SHAPES
    first = True
    for (M, N, K) in shapes:
        A = (torch.rand((M * K,), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        B = (torch.rand((K * N,), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
        D = torch.empty((M * N,), device="cuda", dtype=torch.bfloat16)
        extA = [M, K] if mA == "mk" else [K, M]
        extB = [K, N] if mB == "kn" else [N, K]
        p = ops.contraction_plan(h, extA, mA, extB, mB, [M, N], "mn", dtype=ct.R_16BF, workspace_limit=1 << 30)
        ws = torch.empty(max(p.required_workspace, 16), dtype=torch.uint8, device="cuda")
        if args.beta != 0.0:
            D.zero_()
        fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), args.beta, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace, stream=stream)  # noqa: E731
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        if first:                                  # clock ramp: the first shape of a process would be timed on an idle chip's clocks
            import time
            t0 = time.time()
            while time.time() - t0 < 0.05:
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
            first = False
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        d = p.describe()
        print(json.dumps({"M": M, "N": N, "K": K, "layout": args.layout, "ms": round(ms, 4), "tflops": round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1),
                          "kname": d["kname"], "splitK": d["splitK"], "beta": args.beta}), flush=True)
        p.destroy()
        del A, B, D, ws


if __name__ == "__main__":
    main()
