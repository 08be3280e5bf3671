#!/usr/bin/env python3
"""Randomised parity sweep of the bandwidth-bound operations — cutensorPermute, cutensorReduce,
cutensorElementwiseBinaryExecute, cutensorElementwiseTrinaryExecute — against torch in fp64 on the GPU: random mode
counts, extents (lane-aligned and not), permutations, kept / reduced / broadcast mode subsets, operators, scalars and
data types.  Not part of the test suite; run on a GPU box to look for rare shape bugs (tools/fuzz_contraction.py is the
counterpart for cutensorContract)."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

EXTENTS = [1, 2, 3, 4, 5, 8, 12, 16, 17, 32, 48, 64, 100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--wide", action="store_true", help="extents that are multiples of 64 / 128 / 256 (and near misses): the wide transposing tiles, "
                                                        "their edge tiles and the rest-first tile order")
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    rnd = random.Random(args.seed)
    h = ops.Handle()
    DT = {"float32": (ct.R_32F, 2e-6), "float64": (ct.R_64F, 1e-13), "bfloat16": (ct.R_16BF, 1.2e-2), "float16": (ct.R_16F, 1.5e-3),
          "complex64": (ct.C_32F, 4e-6), "complex128": (ct.C_64F, 2e-13)}     # round 5: (re, im)-pair kernels, ADD / MUL, complex scalars
    fails, done, refused = 0, {}, 0

    def shape(m, ext):
        return [ext[c] for c in m][::-1]          # first listed mode is the fastest: torch's last dimension

    def as_modes(t, m_have, m_want, ext):
        """torch view of tensor t (dims = reversed(m_have)) laid out with dims reversed(m_want); modes of m_want that
        m_have lacks are broadcast."""
        have = list(m_have)[::-1]
        t2 = t
        for c in m_want:
            if c not in have:
                t2 = t2.unsqueeze(0)
                have.insert(0, c)
        perm = [have.index(c) for c in list(m_want)[::-1]]
        return t2.permute(perm).expand(shape(m_want, ext))

    def rand(m, ext, tdt):
        r = torch.rand(shape(m, ext) or [1], device="cuda", dtype=torch.float64) * 2 - 1
        if tdt.is_complex:
            r = torch.complex(r, torch.rand(shape(m, ext) or [1], device="cuda", dtype=torch.float64) * 2 - 1)
        return r.to(tdt).reshape(shape(m, ext))

    def wide(t):
        return t.to(torch.complex128) if t.is_complex() else t.double()

    def rand_strided(m, ext, tdt):
        """(view, column-major element strides or None): sometimes a view into a buffer with padded extents."""
        e = [ext[c] for c in m]
        pad = [rnd.choice([0, 0, 1, 3, 4, 8]) if rnd.random() < 0.4 else 0 for _ in e]
        if not e or not any(pad):
            return rand(m, ext, tdt), None
        full = [x + p_ for x, p_ in zip(e, pad)]
        base = torch.rand(full[::-1], device="cuda", dtype=torch.float64) * 2 - 1
        if tdt.is_complex:
            base = torch.complex(base, torch.rand(full[::-1], device="cuda", dtype=torch.float64) * 2 - 1)
        base = base.to(tdt)
        strides, run = [], 1
        for x in full:
            strides.append(run)
            run *= x
        return base[tuple(slice(0, x) for x in e[::-1])], strides

    def check(kind, got, ref, tol, scale, what):
        nonlocal fails
        err = float((wide(got) - ref).abs().max()) if ref.numel() else 0.0
        bound = tol * max(1.0, scale)
        if not err <= bound:
            fails += 1
            print("MISMATCH %s: err %.3e > %.3e  %s" % (kind, err, bound, what))
        done[kind] = done.get(kind, 0) + 1

    for case in range(args.cases):
        kind = rnd.choice(["permute", "permute", "reduce", "reduce", "binary", "trinary"])
        dtype = rnd.choice(["float32", "float32", "float64", "bfloat16", "float16", "complex64", "complex128"])
        if kind == "trinary" and dtype.startswith("complex"):
            dtype = "float32"                      # complex trinary operations are refused
        cdt, tol = DT[dtype]
        tdt = getattr(torch, dtype)
        cplx = dtype.startswith("complex")
        n = rnd.randint(1, 5)
        labels = rnd.sample("abcdefgh", n)
        ext = {c: rnd.choice([64, 128, 192, 256, 320, 384, 512, 72, 136, 260, 2, 3, 5] if args.wide else EXTENTS) for c in labels}
        vol = 1
        for c in labels:
            vol *= ext[c]
        if vol > (1 << (25 if args.wide else 22)):
            continue
        alpha, gamma, beta = rnd.choice([1.0, 0.5, -1.25]), rnd.choice([0.0, 1.0, -0.5]), rnd.choice([1.0, 0.25])
        if cplx:
            alpha, gamma, beta = alpha + rnd.choice([0.0, 0.75j]), gamma + rnd.choice([0.0, -0.5j]), beta + rnd.choice([0.0, 0.25j])
        try:
            if kind == "permute":
                mA = "".join(labels)
                mB = "".join(rnd.sample(labels, n))
                (A, sA), (B, sB) = rand_strided(mA, ext, tdt), rand_strided(mB, ext, tdt)
                align = rnd.choice([128, 128, 16, A.element_size()])
                plan = ops.permutation_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, dtype=cdt, strideA=sA, strideB=sB,
                                            alignment=align)
                plan.permute(alpha, A.data_ptr(), B.data_ptr())
                torch.cuda.synchronize()
                check(kind, B, alpha * as_modes(wide(A), mA, mB, ext), tol, 1.25,
                      "%s->%s %s %s strides %s %s align %d" % (mA, mB, ext, dtype, sA, sB, align))
            elif kind == "reduce":
                mA = "".join(labels)
                kept = rnd.sample(labels, rnd.randint(0, n - 1)) if n > 1 else []
                mC = "".join(kept)
                op = rnd.choice(["ADD"] if cplx else ["ADD", "ADD", "MAX", "MIN"])
                (A, sA), (D, sC) = rand_strided(mA, ext, tdt), rand_strided(mC, ext, tdt)
                C = D.clone()                      # D aliases C in the call (reduction.cu:219-222); C keeps the input values
                align = rnd.choice([128, 128, 16, A.element_size()])
                plan = ops.reduction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mC], mC, dtype=cdt, op_reduce=ops._OPS[op],
                                          strideA=sA, strideC=sC, alignment=align)
                ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
                plan.reduce(alpha, A.data_ptr(), gamma, D.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
                torch.cuda.synchronize()
                red_dims = [i for i, c in enumerate(list(mA)[::-1]) if c not in kept]
                a64 = wide(A)
                if op == "ADD":
                    r = a64.sum(dim=red_dims) if red_dims else a64
                elif op == "MAX":
                    r = a64.amax(dim=red_dims) if red_dims else a64
                else:
                    r = a64.amin(dim=red_dims) if red_dims else a64
                kept_in_a_order = [c for c in mA if c in kept]
                ref = alpha * as_modes(r.reshape(shape(kept_in_a_order, ext)), "".join(kept_in_a_order), mC, ext) + gamma * wide(C)
                volC = 1
                for c in mC:
                    volC *= ext[c]
                nred = vol // volC
                check(kind, D, ref, tol, 1.25 * (nred ** 0.5 if op == "ADD" else 1.0) + 1.0,
                      "%s->%s %s %s %s alpha %s beta %s strides %s %s align %d" % (mA, mC, op, ext, dtype, alpha, gamma, sA, sC, align))
            elif kind == "binary":
                mC = "".join(labels)
                mA = "".join(rnd.sample(labels, n))
                op = rnd.choice(["ADD", "MUL"] if cplx else ["ADD", "MUL", "MAX", "MIN"])
                A = rand(mA, ext, tdt)
                C = rand(mC, ext, tdt)
                D = torch.empty_like(C)
                plan = ops.binary_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mC], mC, op=op, dtype=cdt)
                plan.binary(alpha, A.data_ptr(), beta, C.data_ptr(), D.data_ptr())
                torch.cuda.synchronize()
                x, y = alpha * as_modes(wide(A), mA, mC, ext), beta * wide(C)
                ref = (x + y) if op == "ADD" else (x * y) if op == "MUL" else torch.maximum(x, y) if op == "MAX" else torch.minimum(x, y)
                check(kind, D, ref, tol, 2.5, "%s,%s %s %s %s" % (mA, mC, op, ext, dtype))
            else:
                mD = "".join(labels)
                mA, mB, mC = ("".join(rnd.sample(labels, n)) for _ in range(3))
                if rnd.random() < 0.4:
                    mC = mD
                opAB, opABC = rnd.choice(["ADD", "MUL", "MAX"]), rnd.choice(["ADD", "MIN", "MUL"])
                A, B, C = rand(mA, ext, tdt), rand(mB, ext, tdt), rand(mC, ext, tdt)
                D = torch.empty(shape(mD, ext), device="cuda", dtype=tdt)
                plan = ops.trinary_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC,
                                        [ext[c] for c in mD], mD, opAB=opAB, opABC=opABC, dtype=cdt)
                plan.trinary(alpha, A.data_ptr(), beta, B.data_ptr(), gamma or 1.0, C.data_ptr(), D.data_ptr())
                torch.cuda.synchronize()
                f = {"ADD": lambda p, q: p + q, "MUL": lambda p, q: p * q, "MAX": torch.maximum, "MIN": torch.minimum}
                x = f[opAB](alpha * as_modes(A.double(), mA, mD, ext), beta * as_modes(B.double(), mB, mD, ext))
                ref = f[opABC](x, (gamma or 1.0) * as_modes(C.double(), mC, mD, ext))
                check(kind, D, ref, tol, 4.0, "%s,%s,%s->%s %s %s %s %s" % (mA, mB, mC, mD, opAB, opABC, ext, dtype))
            plan.destroy()
        except ct.CuTensorError as e:
            refused += 1
            print("case %d: %s refused (%s) %s %s" % (case, kind, e, ext, dtype))
    print("done %s, refused %d, mismatches %d" % (done, refused, fails))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
