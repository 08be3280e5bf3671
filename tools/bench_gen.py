#!/usr/bin/env python3
"""General MFMA family (csrc/kernels/gett_gen.inc) on the shapes the round-3 review names: fp64 4096^3 against the chip's
v_mfma_f64_16x16x4_f64 rate (measured: the guide quotes none), complex64 2048^3 against the fp32 MFMA peak (8 real flop per
complex multiply-add), complex128, the reference's own fp16 case 'mlik,lkjm->lij' at (20,50,50,50) (einsum_test.py:98-107)
and a ragged 16-bit GEMM.  One JSON line per shape; with CUTENSOR_AMD_GEN=0 in the environment the same shapes run on the
kernels they ran on before this family existed (gett_simple_kernel / gett_wide_kernel) — run twice for before / after."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudalibrarysamples_amd import cutensor as ct, ops  # noqa: E402

h = ops.Handle()
FAST = os.environ.get("CUTENSOR_AMD_GEN") == "0"       # the fallback kernels are ~100x slower: fewer repetitions


def time_plan(plan, A, B, D, reps):
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def gemm(name, M, N, K, tdt, cdt, mA="km", mB="kn", flop_per_mac=2.0, reps=10, peak=None):
    eA = [K, M] if mA == "km" else [M, K]
    eB = [K, N] if mB == "kn" else [N, K]
    mk = lambda e: (torch.rand(e[::-1], device="cuda", dtype=torch.float32) * 2 - 1).to(tdt) if not tdt.is_complex else \
        torch.complex(torch.rand(e[::-1], device="cuda") * 2 - 1, torch.rand(e[::-1], device="cuda") * 2 - 1).to(tdt)   # noqa: E731
    A, B = mk(eA), mk(eB)
    D = torch.zeros((N, M), device="cuda", dtype=tdt)
    plan = ops.contraction_plan(h, eA, mA, eB, mB, [M, N], "mn", dtype=cdt, workspace_limit=1 << 30)
    d = plan.describe()
    ms = time_plan(plan, A, B, D, 1 if FAST else reps)
    tf = flop_per_mac * M * N * K / (ms * 1e-3) / 1e12
    out = {"shape": name, "M": M, "N": N, "K": K, "layout": mA + "," + mB, "ms": round(ms, 4), "tflops": round(tf, 2), "kname": d["kname"],
           "tile": [d["bm"], d["bn"], d["bk"]], "vec": d.get("vec"), "splitK": d["splitK"]}
    if peak:
        out["frac_of_peak"] = round(tf / peak, 4)
        out["peak_tflops"] = round(peak, 1)
    print(json.dumps(out), flush=True)
    plan.destroy()


def einsum_case(equation, a_size, b_size, tdt, reps=50):
    from cudalibrarysamples_amd import torch_einsum
    mk = (lambda sz: torch.randn(*sz, device="cuda").to(tdt)) if not tdt.is_complex else \
        (lambda sz: torch.complex(torch.randn(*sz, device="cuda"), torch.randn(*sz, device="cuda")).to(tdt))
    a, b = mk(a_size), mk(b_size)
    out = torch_einsum.einsum(equation, a, b)
    torch.cuda.synchronize()
    p = torch_einsum._plans[(equation, tuple(a.shape), tuple(b.shape), a.dtype, False, False)]
    d = p.describe()
    ws = torch_einsum._get_workspace(a.device, p.required_workspace)
    n = 3 if FAST else reps
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            p.execute(a, b, out, ws)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    wide = torch.complex128 if tdt.is_complex else torch.float64
    ref = torch.einsum(equation, a.to(wide), b.to(wide))
    err = float((out.to(wide) - ref).abs().max() / ref.abs().max())
    print(json.dumps({"shape": "einsum " + equation, "a": list(a_size), "b": list(b_size), "dtype": str(tdt), "us_per_call": round(best * 1e3, 2),
                      "kname": d["kname"], "tile": [d["bm"], d["bn"], d["bk"]], "vec": d.get("vec"), "blocks": d["blocks"], "splitK": d["splitK"],
                      "max_err_over_max_ref": err}), flush=True)


def main():
    which = sys.argv[1:] or ["f64", "c64", "c128", "h16", "einsum"]
    ct.lib.ctamdMeasureMfmaCeilingF64.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    peak64 = None
    if "f64" in which or "c128" in which:
        v0, v1 = ctypes.c_float(0), ctypes.c_float(0)
        ct.lib.ctamdMeasureMfmaCeilingF64(0, ctypes.byref(v0))
        ct.lib.ctamdMeasureMfmaCeilingF64(1, ctypes.byref(v1))
        peak64 = v0.value
        print(json.dumps({"mfma_f64_only_tflops": {"zeros": round(v0.value, 2), "uniform": round(v1.value, 2)}}), flush=True)
    if "f64" in which:
        for (mA, mB) in (("km", "kn"), ("mk", "kn"), ("mk", "nk")):
            gemm("fp64 4096^3", 4096, 4096, 4096, torch.float64, ct.R_64F, mA, mB, reps=5, peak=peak64)
        gemm("fp64 2048^3", 2048, 2048, 2048, torch.float64, ct.R_64F, reps=10, peak=peak64)
        gemm("fp64 1000x1000x1000", 1000, 1000, 1000, torch.float64, ct.R_64F, reps=20, peak=peak64)
    if "c64" in which:
        for (mA, mB) in (("km", "kn"), ("mk", "kn")):
            gemm("complex64 2048^3", 2048, 2048, 2048, torch.complex64, ct.C_32F, mA, mB, flop_per_mac=8.0, reps=10, peak=157.3)
        gemm("complex64 4096^3", 4096, 4096, 4096, torch.complex64, ct.C_32F, flop_per_mac=8.0, reps=3, peak=157.3)
    if "c128" in which:
        gemm("complex128 2048^3", 2048, 2048, 2048, torch.complex128, ct.C_64F, flop_per_mac=8.0, reps=3, peak=peak64)
    if "h16" in which:
        for (mA, mB) in (("km", "kn"), ("mk", "kn"), ("mk", "nk")):
            gemm("bf16 4096x4096x4104 (ragged K)", 4096, 4096, 4104, torch.bfloat16, ct.R_16BF, mA, mB, reps=10, peak=2516.6)
        gemm("bf16 2000^3 (pairs... 16-byte lanes, ragged)", 2000, 2000, 2000, torch.bfloat16, ct.R_16BF, reps=20, peak=2516.6)
        gemm("fp16 1001^3 (2-byte gathers)", 1001, 1001, 1001, torch.float16, ct.R_16F, reps=10, peak=2516.6)
    if "einsum" in which:
        einsum_case("mlik,lkjm->lij", (20, 50, 50, 50), (50, 50, 50, 20), torch.float16)
        einsum_case("mlik,lkjm->lij", (20, 50, 50, 50), (50, 50, 50, 20), torch.bfloat16)
        einsum_case("lik,lkj->lij", (50, 50, 50), (50, 50, 50), torch.float16)
        einsum_case("ik,kj->ij", (50, 50), (50, 50), torch.float16)
        einsum_case("lik,lkj->lij", (50, 50, 50), (50, 50, 50), torch.complex128)
        einsum_case("mlik,lkjm", (2, 5, 50, 2), (5, 2, 50, 2), torch.float64)
        # the headline equation with other element types: one 96 x 96 output tile, K = 262144 -> split-K of the general family
        einsum_case("abcd,dcbe->ae", (96, 64, 64, 64), (64, 64, 64, 96), torch.float64, reps=5)
        einsum_case("abcd,dcbe->ae", (96, 64, 64, 64), (64, 64, 64, 96), torch.complex64, reps=5)
        einsum_case("abcd,dcbe->ae", (96, 64, 64, 64), (64, 64, 64, 96), torch.bfloat16, reps=20)


if __name__ == "__main__":
    main()
