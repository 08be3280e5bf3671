#!/usr/bin/env python3
"""Shapes a user of an einsum library brings that are NOT the GEMM-like benchmark shapes: batched contractions with a short contracted
range (attention scores / values), skinny outputs, many small batches.  Each through torch_einsum.einsum (the C ABI, planner's choice)
beside torch.einsum (the vendor BLAS) on the same tensors: us per call, TFLOP/s, the kernel the planner took.  One JSON line per shape."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [
    ("bhqd,bhkd->bhqk", dict(b=8, h=8, q=2048, k=2048, d=128), "attention scores: K = 128, 64 batches, 537 MB of output"),
    ("bhqk,bhkd->bhqd", dict(b=8, h=8, q=2048, k=2048, d=128), "attention values: N = 128"),
    ("bhqd,bhkd->bhqk", dict(b=4, h=16, q=1024, k=1024, d=64), "attention scores: K = 64, one K-tile"),
    ("bij,bjk->bik", dict(b=64, i=1024, j=1024, k=1024), "64 batches of 1024^3"),
    ("bij,bjk->bik", dict(b=512, i=256, j=256, k=256), "512 batches of 256^3"),
    ("ik,kj->ij", dict(i=16384, j=16384, k=128), "flat, K = 128: 537 MB of output"),
    ("ik,kj->ij", dict(i=8192, j=128, k=8192), "skinny N = 128"),
    ("bik,bjk->bij", dict(b=32, i=2048, j=2048, k=256), "both K-contiguous, 4 K-tiles, batched"),
]

# EINSUM_SHAPES_SET=sweep: several contracted modes that do not fuse, the fastest one without whole 64-deep K-tiles (round 6: the masked
# last K-tile of every sweep keeps them on the LDS-DMA kernels; CUTENSOR_AMD_GEN=f in the hooks flavour shows the general family beside it)
SWEEP_CASES = [
    ("abcd,dcbe->ae", dict(a=2048, b=8, c=8, d=96, e=2048), "the headline equation, contracted extents 8 x 8 x 96 (two K-tiles per sweep, 75 % live)"),
    ("abcd,dcbe->ae", dict(a=2048, b=8, c=16, d=40, e=2048), "contracted extents 8 x 16 x 40 (one K-tile per sweep, 62 % live)"),
    ("abcd,dcbe->ae", dict(a=4096, b=4, c=8, d=200, e=4096), "contracted extents 4 x 8 x 200 (four K-tiles per sweep, 78 % live)"),
    ("ijk,lkj->il", dict(i=4096, l=4096, j=16, k=72), "A contiguous in k, B in j"),
    ("abcd,dcbe->ae", dict(a=96, b=96, c=96, d=96, e=96), "the headline equation at extents of 96 (split-K)"),
    # operands the LDS-DMA kernels cannot stage as they lie: copied into packed temporaries first (api.cpp plan_repack)
    ("abcd,dcbe->ae", dict(a=2048, b=4, c=16, d=50, e=2048), "contracted extents 4 x 16 x 50: partial 16-byte units at the end of every sweep"),
    ("abcd,dcbe->ae", dict(a=2048, b=8, c=16, d=16, e=2048), "contracted extents 8 x 16 x 16: a quarter of every K-tile live"),
    ("mlik,lkjm->lij", dict(m=64, l=64, i=512, k=64, j=512), "the reference's test equation, larger: A contiguous in k, B in m"),
]
if os.environ.get("EINSUM_SHAPES_SET") == "sweep":
    CASES = SWEEP_CASES


def timed(torch, fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def flop_of(ext, a_m, b_m):
    f = 2.0
    for c in set(a_m + b_m):
        f *= ext[c]
    return f


def main():
    import torch
    from cudalibrarysamples_amd import torch_einsum
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f64": torch.float64, "c64": torch.complex64}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
    only = {int(x) for x in os.environ.get("EINSUM_SHAPES_ONLY", "").split(",") if x}   # indices into CASES (empty: all)
    for idx, (eq, ext, note) in enumerate(CASES):
        if only and idx not in only:
            continue
        ins, out = eq.split("->")
        a_m, b_m = ins.split(",")
        if dt == torch.float32 and flop_of(ext, a_m, b_m) > 4e11:
            continue                                   # (the largest cases are bf16-sized)
        a = (torch.rand([ext[c] for c in a_m], device="cuda") * 2 - 1).to(dt)
        b = (torch.rand([ext[c] for c in b_m], device="cuda") * 2 - 1).to(dt)
        flop = 2.0
        for c in set(a_m + b_m):
            flop *= ext[c]
        res = torch_einsum.einsum(eq, a, b)
        ref = torch.einsum(eq, a, b)
        err = float((res.float() - ref.float()).abs().max() / ref.float().abs().max())
        p = torch_einsum._plans[(eq, tuple(a.shape), tuple(b.shape), a.dtype, False, False)]
        d = p.describe()
        d["kname"] = ("copy " + "AB"[0:d["repack_A"]] + "AB"[1:1 + d["repack_B"]] + " + " if d.get("repack_A") or d.get("repack_B") else "") + str(d.get("kname"))
        ws = torch_einsum._get_workspace(a.device, p.required_workspace)
        o = torch.empty_like(res)
        ms = timed(torch, lambda: p.execute(a, b, o, ws), 20)
        ms_v = timed(torch, lambda: torch.einsum(eq, a, b), 20)
        nbytes = float(a.element_size()) * (a.numel() + b.numel() + res.numel())
        print(json.dumps({"dtype": str(dt), "equation": eq, "extents": ext, "note": note, "us": round(ms * 1e3, 1), "tflops": round(flop / (ms * 1e-3) / 1e12, 1),
                          "GBps_algorithmic": round(nbytes / (ms * 1e-3) / 1e9), "vendor_us": round(ms_v * 1e3, 1),
                          "vendor_tflops": round(flop / (ms_v * 1e-3) / 1e12, 1), "kernel": d.get("kname"), "tile": [d.get("bm"), d.get("bn"), d.get("bk")],
                          "splitK": d.get("splitK"), "blocks": d.get("blocks"), "max_rel_diff_vs_vendor": err}), flush=True)
        del a, b, res, ref, o


if __name__ == "__main__":
    main()
