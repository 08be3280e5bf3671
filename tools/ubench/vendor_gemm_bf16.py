#!/usr/bin/env python3
"""Calibration point for BASELINE configs[3], NOT part of the engine: the vendor GEMM (torch.matmul -> hipBLASLt /
rocBLAS) on the same box, the same 8192^3 bf16 shape, the same U(-1,1) data and the same steady-state protocol as
tools/bench_h16.py.  The bf16 MFMA rate on MI355X is data-dependent (power): a fraction of the 2.52 PFLOP/s peak
only means something next to what the vendor library reaches on identical operands."""
import argparse
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--zeros", action="store_true")
    args = ap.parse_args()
    import torch
    n = args.n
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    A = (torch.rand((n, n), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((n, n), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    if args.zeros:
        A.zero_()
        B.zero_()
    out = {}
    for name, fn in (("nn", lambda: torch.matmul(A, B)), ("nt", lambda: torch.matmul(A, B.t())), ("tn", lambda: torch.matmul(A.t(), B))):
        D = torch.empty((n, n), device="cuda", dtype=torch.bfloat16)
        ops = {"nn": (A, B), "nt": (A, B.t()), "tn": (A.t(), B)}[name]
        for _ in range(60):
            torch.matmul(ops[0], ops[1], out=D)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            torch.matmul(ops[0], ops[1], out=D)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        out[name] = {"ms": round(ms, 4), "tflops": round(2.0 * n ** 3 / ms / 1e9, 1)}
    print(json.dumps({"what": "torch.matmul bf16 %d^3 (vendor GEMM calibration)" % n, "zeros": args.zeros, "layouts": out,
                      "frac_of_2.52PF": {k: round(v["tflops"] / 2516.6, 3) for k, v in out.items()}}))


if __name__ == "__main__":
    main()
