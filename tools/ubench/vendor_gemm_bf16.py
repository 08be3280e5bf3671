#!/usr/bin/env python3
"""Calibration point for BASELINE configs[3], NOT part of the engine: the vendor GEMM (torch.matmul -> hipBLASLt /
rocBLAS) on the same box, the same 8192^3 bf16 shape, the same U(-1,1) data and the same steady-state protocol as
tools/bench_h16.py.  The bf16 MFMA rate on MI355X is data-dependent (power): a fraction of the 2.52 PFLOP/s peak
only means something next to what the vendor library reaches on identical operands."""
import argparse
import json


def run_shape(torch, M, N, K, reps, zeros):
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    mk = lambda r, c: (torch.zeros((r, c), device="cuda", dtype=torch.bfloat16) if zeros
                       else (torch.rand((r, c), generator=g, device="cuda") * 2 - 1).to(torch.bfloat16))   # noqa: E731
    A, At, B, Bt = mk(M, K), mk(K, M), mk(K, N), mk(N, K)
    D = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    out = {}
    for name, ops in (("nn", (A, B)), ("nt", (A, Bt.t())), ("tn", (At.t(), B))):
        warm = max(20, min(60, int(30e-3 / max(2.0 * M * N * K / 1.5e15, 1e-6))))
        for _ in range(warm):
            torch.matmul(ops[0], ops[1], out=D)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.matmul(ops[0], ops[1], out=D)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": round(ms, 5), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--shapes", default="", help="M,N,K;M,N,K;... (round 5: the mid-size yardstick, one JSON line per shape)")
    args = ap.parse_args()
    import torch
    if args.shapes:
        for sh in args.shapes.split(";"):
            M, N, K = (int(x) for x in sh.split(","))
            out = run_shape(torch, M, N, K, args.reps, args.zeros)
            print(json.dumps({"what": "torch.matmul bf16 (vendor GEMM calibration)", "M": M, "N": N, "K": K, "zeros": args.zeros, "layouts": out,
                              "best_tflops": max(v["tflops"] for v in out.values())}), flush=True)
        return
    n = args.n
    out = run_shape(torch, n, n, n, args.reps, args.zeros)
    print(json.dumps({"what": "torch.matmul bf16 %d^3 (vendor GEMM calibration)" % n, "zeros": args.zeros, "layouts": out,
                      "frac_of_2.52PF": {k: round(v["tflops"] / 2516.6, 3) for k, v in out.items()}}))


if __name__ == "__main__":
    main()
