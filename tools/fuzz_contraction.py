#!/usr/bin/env python3
"""Randomised parity sweep of cutensorContract (fp32 and bf16) against torch.einsum in fp64 on the GPU: random mode
counts, extents (biased to the 16-byte-lane / K-tile conditions of the MFMA kernels, but not only), mode orders,
alpha / beta, workspace limits.  Not part of the test suite; run on a GPU box to look for rare shape bugs."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--aligned", action="store_true", help="extents that satisfy the 16-byte-lane / whole-K-tile conditions of the "
                                                           "streaming fp32 and the 16-bit MFMA kernels")
    ap.add_argument("--all-types", action="store_true", help="also fp64, fp16, complex<float> and complex<double> (the general MFMA family)")
    ap.add_argument("--strided", action="store_true", help="give a third of the tensors padded (non-packed) strides")
    ap.add_argument("--many-modes", action="store_true", help="3-6 small modes per group: groups beyond the tiled kernels' four digits "
                                                              "(peeled into a host loop up to 64 launches, mode-table kernel beyond)")
    ap.add_argument("--ragged-k", action="store_true", help="16-bit data, 16-byte lanes, ONE contracted mode whose extent is a multiple of 8 "
                                                            "but not of 64: the masked last K-tile of the LDS-DMA 16-bit kernels")
    ap.add_argument("--sweep-k", action="store_true", help="16-bit data, two or three contracted modes with extents that are multiples of 8 but "
                                                           "mostly not of 64, free extents multiples of 8: the sweep mask of the LDS-DMA kernels "
                                                           "(and, with CUTENSOR_AMD_REPACK=f in the hooks flavour, operands copied into temporaries first)")
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    rnd = random.Random(args.seed)
    h = ops.Handle()
    fails = 0
    kinds = {}
    for case in range(args.cases):
        dtype = rnd.choice(["float32", "float32", "bfloat16"] + (["float64", "float16", "complex64", "complex128"] if args.all_types else []))
        nM, nN, nK, nL = rnd.randint(1, 2), rnd.randint(1, 2), rnd.randint(1, 3), rnd.choice([0, 0, 0, 1])
        if args.ragged_k:
            dtype, nK = rnd.choice(["bfloat16", "bfloat16", "float16"]), 1
        if args.sweep_k:
            dtype, nK = rnd.choice(["bfloat16", "bfloat16", "float16"]), rnd.randint(2, 3)
        if args.many_modes:
            nM, nN, nK, nL = rnd.randint(2, 6), rnd.randint(1, 5), rnd.randint(1, 6), rnd.choice([0, 0, 1, 2])
        labels = list("abcdefghijklmnopqrstuvwxyz")
        rnd.shuffle(labels)
        M, N, K, L = [labels.pop() for _ in range(nM)], [labels.pop() for _ in range(nN)], [labels.pop() for _ in range(nK)], [labels.pop() for _ in range(nL)]
        ext = {}
        table = ((M, [8, 16, 24, 40, 96, 100, 7]), (N, [8, 16, 32, 48, 96, 13]), (K, [4, 8, 32, 64, 64, 96, 5]), (L, [2, 3]))
        if args.aligned:
            table = ((M, [8, 16, 24, 40, 96, 104, 264]), (N, [8, 16, 32, 48, 96, 120]), (K, [64, 64, 128, 192]), (L, [2, 3]))
        if args.many_modes:
            table = ((M, [2, 3, 4, 5, 8]), (N, [2, 3, 4, 6, 8]), (K, [2, 3, 4, 8]), (L, [2, 3]))
        if args.ragged_k:
            table = ((M, [8, 16, 24, 40, 96, 104, 264, 520]), (N, [8, 16, 32, 48, 96, 120, 392]), (K, [8, 24, 72, 136, 200, 328, 520, 1000, 2056, 4104]), (L, [2, 3]))
        if args.sweep_k:
            table = ((M, [8, 16, 24, 40, 96, 104, 264, 520, 37, 100]), (N, [8, 16, 32, 48, 96, 120, 392, 29, 130]), (K, [2, 3, 8, 16, 24, 40, 50, 64, 72, 96, 136, 200]), (L, [2, 3]))
        for g, choices in table:
            for c in g:
                ext[c] = rnd.choice(choices)
        mA, mB, mC = M + K + L, N + K + L, M + N + L
        for m in (mA, mB, mC):
            rnd.shuffle(m)
        mA, mB, mC = "".join(mA), "".join(mB), "".join(mC)
        elems = lambda m: eval("*".join(str(ext[c]) for c in m))
        if max(elems(mA), elems(mB), elems(mC)) > (1 << 24):
            continue
        tdt = getattr(torch, dtype)

        def make(m):
            """Tensor with modes m (first fastest); with --strided sometimes a view into a buffer with padded extents.
            Returns (torch view, column-major element strides or None)."""
            e = [ext[c] for c in m]
            pad = [rnd.choice([0, 0, 1, 3, 8]) if (args.strided and rnd.random() < 0.35) else 0 for _ in e]
            full = [x + p_ for x, p_ in zip(e, pad)]
            if tdt.is_complex:
                base = torch.complex(torch.rand(full[::-1], device="cuda") * 2 - 1, torch.rand(full[::-1], device="cuda") * 2 - 1).to(tdt)
            else:
                base = (torch.rand(full[::-1], device="cuda") * 2 - 1).to(tdt)
            view = base[tuple(slice(0, x) for x in e[::-1])]
            if not any(pad):
                return view, None
            strides, run = [], 1
            for x in full:
                strides.append(run)
                run *= x
            return view, strides

        (A, sA), (B, sB), (C, sC) = make(mA), make(mB), make(mC)
        D = C.clone() if sC is None else C          # strided output: written in place (C aliases D, contraction.cu:264)
        C0 = C.clone()
        alpha, beta = rnd.choice([1.0, 0.5, -1.25]), rnd.choice([0.0, 0.0, 1.0, -0.5])
        limit = rnd.choice([0, 1 << 20, 1 << 28])
        try:
            cdt = {"float32": ct.R_32F, "bfloat16": ct.R_16BF, "float64": ct.R_64F, "float16": ct.R_16F, "complex64": ct.C_32F, "complex128": ct.C_64F}[dtype]
            plan = ops.contraction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC,
                                        dtype=cdt, workspace_limit=limit, strideA=sA, strideB=sB, strideC=sC)
        except ct.CuTensorError as e:
            print("case %d: plan refused (%s) %s,%s->%s %s" % (case, e, mA, mB, mC, ext))
            fails += 1
            continue
        d = plan.describe()
        kname = d.get("kname") + ("+peel" if d.get("peeled_modes") else "") + ("+copy" if d.get("repack_A") or d.get("repack_B") else "") + \
            ("+sweep" if d.get("rag") and len(d.get("Kdigits", [])) > 1 else "")
        kinds[(kname, d["splitK"] > 1)] = kinds.get((kname, d["splitK"] > 1), 0) + 1
        ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
        plan.contract(alpha, A.data_ptr(), B.data_ptr(), beta, C.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
        torch.cuda.synchronize()
        wide = torch.complex128 if tdt.is_complex else torch.float64
        ref = alpha * torch.einsum("%s,%s->%s" % (mA[::-1], mB[::-1], mC[::-1]), A.to(wide), B.to(wide)) + beta * C0.to(wide)
        err = (D.to(wide) - ref).abs()
        ktot = 1
        for c in K:
            ktot *= ext[c]
        rel, absk = {"float32": (2e-5, 1e-5), "complex64": (4e-5, 2e-5), "float64": (1e-12, 1e-13), "complex128": (2e-12, 2e-13), "float16": (2e-3, 1e-3),
                     "bfloat16": (1e-2, 4e-3)}[dtype]
        # sequential accumulation: ~eps * sqrt(K) steps, each relative to partial sums that themselves grow like sqrt(K)
        tol = rel * (1.0 + ref.abs()) + absk * ktot ** 0.5 + rel * 1e-3 * ktot
        if not bool((err <= tol).all()):
            fails += 1
            print("case %d MISMATCH max err %.3e: %s,%s->%s %s %s alpha %g beta %g limit %d plan %s" % (
                case, float(err.max()), mA, mB, mC, ext, dtype, alpha, beta, limit, d))
        plan.destroy()
    print("cases %d, failures %d, kernels used: %s" % (args.cases, fails, kinds))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
