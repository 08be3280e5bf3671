#!/usr/bin/env python3
"""BASELINE configs[2]: cutensorPermute / cutensorReduce on a cubic fp32 tensor, timed with events on the
launch stream; GB/s by the samples' formulas (elementwise_permute.cu:208 = 2*|C| bytes,
reduction.cu:229-231 = |A| + |C| bytes) against the HBM roofline.  Tensors are generated on the device
(never mirror the samples' pinned-host copies at 32 GiB).  One JSON line per case."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_TBPS = 8.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true", help="sampled gather check against torch indexing")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16", "f64", "c64", "c128"], help="element type")
    ap.add_argument("--ext", default="", help="extents a,b,c (default: n,n,n) — e.g. 2048,2048,1024 for the fp64 case of VERDICT r5")
    ap.add_argument("--only", default="", help="comma list of cases, e.g. permute:cab,reduce:ac (default: all) — bench.py's PMC children")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    import torch
    from cudalibrarysamples_amd import ops, cutensor as ct
    n = args.n
    tdt, cdt, es = {"f32": (torch.float32, ct.R_32F, 4), "bf16": (torch.bfloat16, ct.R_16BF, 2), "f16": (torch.float16, ct.R_16F, 2),
                    "f64": (torch.float64, ct.R_64F, 8), "c64": (torch.complex64, ct.C_32F, 8), "c128": (torch.complex128, ct.C_64F, 16)}[args.dtype]
    ea, eb, ec = (int(x) for x in args.ext.split(",")) if args.ext else (n, n, n)
    numel = ea * eb * ec
    free, _ = torch.cuda.mem_get_info()
    need = 2 * numel * max(es, 4) + numel * 4 + (1 << 30)
    if free < need:
        raise SystemExit("not enough free HBM: need %d have %d" % (need, free))
    h = ops.Handle()
    stream = torch.cuda.current_stream().cuda_stream
    # counter-based fill: a cheap hash of the linear index (fixed seed), generated in chunks on the device
    A = torch.empty(numel, dtype=torch.float32, device="cuda")
    chunk = 1 << 28
    for s in range(0, numel, chunk):
        e = min(numel, s + chunk)
        idx = torch.arange(s, e, device="cuda", dtype=torch.int64)
        A[s:e] = ((idx * 2654435761 + 1234) % 16777216).to(torch.float32) / 16777216.0
        del idx
    if tdt in (torch.complex64, torch.complex128):
        re = A.to(torch.float32 if tdt == torch.complex64 else torch.float64)
        A = torch.complex(re, 0.5 - re)
        del re
    elif tdt != torch.float32:
        A = A.to(tdt)
    ext = dict(a=ea, b=eb, c=ec)

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    # ---- permutations (column-major mode lists: first mode is stride-1) ------------------------------
    for mB in ("cab", "cba", "acb"):
        if only and "permute:" + mB not in only:
            continue
        D = torch.empty(numel, dtype=tdt, device="cuda")
        p = ops.permutation_plan(h, [ea, eb, ec], "abc", [ext[c] for c in mB], mB, dtype=cdt)
        ms = timed(lambda: p.permute(1.0, A.data_ptr(), D.data_ptr(), stream))
        gbs = 2.0 * numel * es / (ms * 1e-3) / 1e9
        line = {"op": "permute abc->" + mB, "dtype": args.dtype, "n": n, "ext": [ea, eb, ec], "ms": ms, "GBps": gbs, "frac_hbm_peak": gbs / (HBM_PEAK_TBPS * 1e3),
                "plan": p.describe()}
        if args.check:
            At = A.view(ec, eb, ea)    # At[c][b][a] (row-major view of the column-major tensor)
            Dt = D.view(ext[mB[2]], ext[mB[1]], ext[mB[0]])    # Dt[m2][m1][m0] with modes mB = m0 m1 m2
            g = torch.Generator(device="cuda"); g.manual_seed(7)
            ia, ib, ic = (torch.randint(0, e, (1 << 16,), generator=g, device="cuda") for e in (ea, eb, ec))
            pos = dict(a=ia, b=ib, c=ic)
            got = Dt[pos[mB[2]], pos[mB[1]], pos[mB[0]]]
            line["sampled_mismatches"] = int((got != At[ic, ib, ia]).sum().item())
        print(json.dumps(line), flush=True)
        p.destroy()
        del D
    # ---- reductions -------------------------------------------------------------------------------------
    for mC in ("ac", "c", "a", "bc"):
        if only and "reduce:" + mC not in only:
            continue
        eC = [ext[c] for c in mC]
        outn = int(np.prod(eC))
        D = torch.zeros(outn, dtype=tdt, device="cuda")
        p = ops.reduction_plan(h, [ea, eb, ec], "abc", eC, mC, dtype=cdt, workspace_limit=1 << 30)
        ws = torch.empty(max(p.required_workspace, 256), dtype=torch.uint8, device="cuda")
        ms = timed(lambda: p.reduce(1.1, A.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace, stream))
        gbs = (numel + outn) * float(es) / (ms * 1e-3) / 1e9
        line = {"op": "reduce abc->" + mC, "dtype": args.dtype, "n": n, "ext": [ea, eb, ec], "ms": ms, "GBps": gbs, "frac_hbm_peak": gbs / (HBM_PEAK_TBPS * 1e3),
                "plan": p.describe()}
        if args.check:
            dims = tuple(i for i, c in enumerate("cba") if c not in mC)
            wide = torch.complex128 if tdt.is_complex else torch.float64
            ref = (A.view(ec, eb, ea).sum(dim=dims, dtype=wide) * 1.1)
            # ref is indexed in row-major order of the kept modes in 'cba' order == column-major mC order reversed
            got = D.view(*[ext[c] for c in reversed(mC)]).to(wide)
            line["max_rel_err"] = float(((got - ref).abs() / ref.abs().clamp_min(1e-30)).max().item())
        print(json.dumps(line), flush=True)
        p.destroy()
        del D, ws


if __name__ == "__main__":
    main()
