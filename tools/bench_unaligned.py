#!/usr/bin/env python3
"""Rates of 16-bit / fp32 contractions whose operands have no 16-byte lanes (Missing #1): C[m,n] = sum_k A[m,k] B[k,n] with
extents like 4100 (every row 8 (mod 16) bytes), 4097 (odd pitch) or 4098 (fp32: 8 (mod 16)), on the four operand layouts, through
cutensorContract — next to the aligned neighbour (4096) and, with --vendor, the vendor GEMM (torch.matmul: hipBLASLt / rocBLAS) on the
same box as a yardstick (never linked by the engine).  One JSON line per (shape, layout)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LAYOUTS = (("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk"))


def time_calls(torch, fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096,4096,4096;4100,4100,4100;4097,4097,4097;4104,4104,4104")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--warm", type=int, default=30)
    ap.add_argument("--vendor", action="store_true")
    ap.add_argument("--layouts", default="all")
    ap.add_argument("--beta", type=float, default=0.0)
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    tdt = getattr(torch, args.dtype)
    cdt = {"bfloat16": ct.R_16BF, "float16": ct.R_16F, "float32": ct.R_32F}[args.dtype]
    h = ops.Handle()
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    stream = torch.cuda.current_stream().cuda_stream
    layouts = LAYOUTS if args.layouts == "all" else tuple(tuple(x.split(",")) for x in args.layouts.split(";"))
    for shape in args.shapes.split(";"):
        M, N, K = (int(x) for x in shape.split(","))
        ext = dict(m=M, n=N, k=K)
        flop = 2.0 * M * N * K
        for (mA, mB) in layouts:
            eA, eB = [ext[c] for c in mA], [ext[c] for c in mB]
            A = (torch.rand(eA[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
            B = (torch.rand(eB[::-1], generator=g, device="cuda") * 2 - 1).to(tdt)
            D = torch.zeros((N, M), device="cuda", dtype=tdt)
            plan = ops.contraction_plan(h, eA, mA, eB, mB, [M, N], "mn", dtype=cdt, workspace_limit=1 << 30)
            d = plan.describe()
            ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
            run = lambda: plan.contract(1.0, A.data_ptr(), B.data_ptr(), args.beta, D.data_ptr(), D.data_ptr(), ws.data_ptr(),   # noqa: E731
                                        plan.required_workspace, stream=stream)
            ms = time_calls(torch, run, args.reps, args.warm)
            line = {"M": M, "N": N, "K": K, "dtype": args.dtype, "layout": "%s,%s" % (mA, mB), "beta": args.beta, "us": round(ms * 1e3, 2),
                    "tflops": round(flop / ms / 1e9, 1), "family": d["family"], "kname": d["kname"], "splitK": d["splitK"],
                    "strips": d.get("strips", 0), "rag": d.get("rag", 0), "model_us": d["model_us"]}
            if args.vendor:
                # the same product as a vendor GEMM on row-major views: D[n][m] = sum_k Bm[n,k] Am[k,m]
                Am = A if mA == "mk" else A.t()          # [k, m]
                Bm = B.t() if mB == "nk" else B          # kn: tensor [n][k] = [n, k]
                out = torch.empty((N, M), device="cuda", dtype=tdt)
                vms = time_calls(torch, lambda: torch.matmul(Bm, Am, out=out), args.reps, args.warm)
                line["vendor_us"] = round(vms * 1e3, 2)
                line["vendor_tflops"] = round(flop / vms / 1e9, 1)
            print(json.dumps(line), flush=True)
            plan.destroy()
            del A, B, D


if __name__ == "__main__":
    main()
