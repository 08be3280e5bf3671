#!/usr/bin/env python3
"""Randomised parity sweep of the einsum front end (cutensor_amd::Einsum<> behind cudalibrarysamples_amd.torch_einsum,
the counterpart of cuTENSOR/einsum.cu and python/cutensor/torch/einsum.py): random one- and two-operand equations —
explicit and implicit outputs, batch / contracted / free / summed-away modes, scalar results, permutations and
reductions of a single operand — in fp32 / fp64 / fp16 / bf16 / complex64, against torch.einsum in double precision.
Also differentiates a third of the real-valued cases and compares the gradients.  Not part of the test suite."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=300)
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import torch_einsum as te
    rnd = random.Random(args.seed)
    TOL = {torch.float32: 2e-5, torch.float64: 1e-12, torch.float16: 4e-3, torch.bfloat16: 2e-2, torch.complex64: 4e-5}
    fails, refused, kinds = 0, 0, {}
    for case in range(args.cases):
        dtype = rnd.choice([torch.float32, torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.complex64])
        labels = rnd.sample("abcdefghij", rnd.randint(1, 6))
        ext = {c: rnd.choice([1, 2, 3, 4, 5, 8, 16, 24]) for c in labels}
        binary = rnd.random() < 0.7
        if binary:
            ma = rnd.sample(labels, rnd.randint(1, len(labels)))
            mb = rnd.sample(labels, rnd.randint(1, len(labels)))
            present = [c for c in labels if c in ma or c in mb]
        else:
            ma, mb = rnd.sample(labels, rnd.randint(1, len(labels))), None
            present = list(ma)
        out = rnd.sample(present, rnd.randint(0, len(present)))
        implicit = rnd.random() < 0.25
        lhs = "".join(ma) + ("," + "".join(mb) if binary else "")
        eq = lhs if implicit else lhs + "->" + "".join(out)
        wide = torch.complex128 if dtype.is_complex else torch.float64

        def rand(m):
            shape = [ext[c] for c in m]
            if dtype.is_complex:
                return torch.complex(torch.rand(shape, device="cuda") * 2 - 1, torch.rand(shape, device="cuda") * 2 - 1).to(dtype)
            return (torch.rand(shape, device="cuda", dtype=torch.float64) * 2 - 1).to(dtype)

        a = rand(ma)
        b = rand(mb) if binary else None
        grad = (not dtype.is_complex) and dtype in (torch.float32, torch.float64) and rnd.random() < 0.35
        try:
            if grad:
                a1 = a.clone().requires_grad_(True)
                b1 = b.clone().requires_grad_(True) if binary else None
                got = te.EinsumFunction.apply(eq, a1, b1)
            else:
                got = te.einsum(te.normalize_subscript(eq)[0], a, b)
        except (ValueError, RuntimeError) as e:
            refused += 1
            print("case %d refused: '%s' %s %s: %s" % (case, eq, ext, dtype, str(e)[:120]))
            continue
        ops = [a.to(wide)] + ([b.to(wide)] if binary else [])
        ref = torch.einsum(eq, *ops)
        nsum = 1
        for c in present:
            if c not in (ref.shape and [] or []):
                pass
        kterms = 1
        for c in set(present) - set(te.normalize_subscript(eq)[0].split("->")[1]):
            kterms *= ext[c]
        tol = TOL[dtype] * (1.0 + float(ref.abs().max()) if ref.numel() else 1.0) * max(1.0, kterms ** 0.5)
        ok = tuple(got.shape) == tuple(ref.shape) and (ref.numel() == 0 or float((got.to(wide) - ref).abs().max()) <= tol)
        kind = ("binary" if binary else "unary") + ("/implicit" if implicit else "") + ("/grad" if grad else "")
        kinds[kind] = kinds.get(kind, 0) + 1
        if ok and grad:
            w = torch.rand(ref.shape, device="cuda", dtype=torch.float64) if ref.numel() else torch.zeros(ref.shape, device="cuda", dtype=torch.float64)
            got.backward(w.to(dtype))
            a2 = a.to(wide).requires_grad_(True)
            b2 = b.to(wide).requires_grad_(True) if binary else None
            torch.einsum(eq, *([a2] + ([b2] if binary else []))).backward(w)
            for g1, g2, name in ((a1.grad, a2.grad, "dA"), (b1.grad if binary else None, b2.grad if binary else None, "dB")):
                if g2 is None:
                    continue
                if g1 is None or float((g1.to(wide) - g2).abs().max()) > 50 * tol * (1.0 + float(g2.abs().max())):
                    ok = False
                    print("case %d gradient %s differs: '%s' %s" % (case, name, eq, ext))
        if not ok:
            fails += 1
            err = float((got.to(wide) - ref).abs().max()) if tuple(got.shape) == tuple(ref.shape) and ref.numel() else -1.0
            print("case %d MISMATCH '%s' %s %s: shapes %s vs %s, max err %.3e (tol %.3e)" % (
                case, eq, ext, dtype, tuple(got.shape), tuple(ref.shape), err, tol))
    print("cases %d, refused %d, failures %d, kinds %s" % (args.cases, refused, fails, kinds))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
