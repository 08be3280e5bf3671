#!/usr/bin/env python3
"""Randomised parity sweep aimed at the fp32 LDS-DMA ring kernels (gett_f32_stream.hip — the headline kernel family):
small outputs (one or a few 64/96/128 tiles, ragged edges), long multi-digit contracted modes whose total is a multiple
of the 32-deep K-tile, every operand layout, and for each shape the first ranked candidates (tile / ring depth / split-K
variants) instead of only the planner's favourite.  Checked against torch.einsum in fp64.  Not part of the test suite."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=120)
    ap.add_argument("--ranks", type=int, default=6, help="candidates tried per shape")
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensor as ct, ops
    rnd = random.Random(args.seed)
    h = ops.Handle()
    fails, used, tried = 0, {}, 0
    for case in range(args.cases):
        nK = rnd.randint(1, 3)
        labels = list("abcdefgh")
        rnd.shuffle(labels)
        M, N = [labels.pop()], [labels.pop()]
        if rnd.random() < 0.25:
            M.append(labels.pop())
        K = [labels.pop() for _ in range(nK)]
        ext = {}
        for c in M:
            ext[c] = rnd.choice([8, 16, 24, 48, 64, 96, 100, 128, 136]) if len(M) == 1 else rnd.choice([4, 8, 12])
        for c in N:
            ext[c] = rnd.choice([16, 32, 64, 96, 104, 128, 160])
        kext = {1: [256, 1024, 4096, 8192, 16384], 2: [8, 16, 32, 64, 96], 3: [4, 8, 16, 32]}[nK]
        for c in K:
            ext[c] = rnd.choice(kext)
        ktot = 1
        for c in K:
            ktot *= ext[c]
        if ktot % 32 != 0 or ktot < 128:
            continue
        mA, mB, mC = M + K, N + K, M + N
        for m in (mA, mB):
            rnd.shuffle(m)
        if rnd.random() < 0.5:
            mC = mC[::-1]
        mA, mB, mC = "".join(mA), "".join(mB), "".join(mC)
        A = torch.rand([ext[c] for c in mA][::-1], device="cuda") * 2 - 1
        B = torch.rand([ext[c] for c in mB][::-1], device="cuda") * 2 - 1
        C = torch.rand([ext[c] for c in mC][::-1], device="cuda") * 2 - 1
        alpha, beta = rnd.choice([1.0, -0.75]), rnd.choice([0.0, 0.5])
        ref = alpha * torch.einsum("%s,%s->%s" % (mA[::-1], mB[::-1], mC[::-1]), A.double(), B.double()) + beta * C.double()
        limit = rnd.choice([1 << 30, 1 << 30, 1 << 22, 0])
        for rank in range(args.ranks):
            try:
                plan = ops.contraction_plan(h, [ext[c] for c in mA], mA, [ext[c] for c in mB], mB, [ext[c] for c in mC], mC,
                                            workspace_limit=limit, kernel_rank=rank)
            except ct.CuTensorError:
                break
            d = plan.describe()
            key = (d["kname"], d["bm"], d["pf"], d["splitK"] > 1)
            used[key] = used.get(key, 0) + 1
            tried += 1
            D = C.clone()
            ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
            plan.contract(alpha, A.data_ptr(), B.data_ptr(), beta, D.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
            torch.cuda.synchronize()
            err = (D.double() - ref).abs()
            tol = 2e-5 * (1.0 + ref.abs()) + 1e-5 * ktot ** 0.5 + 2e-8 * ktot
            if not bool((err <= tol).all()):
                fails += 1
                print("case %d rank %d MISMATCH max err %.3e: %s,%s->%s %s alpha %g beta %g limit %d plan %s" % (
                    case, rank, float(err.max()), mA, mB, mC, ext, alpha, beta, limit, d))
            plan.destroy()
    print("plans checked %d, failures %d" % (tried, fails))
    for k in sorted(used, key=lambda x: -used[x]):
        print("  %-26s tile %3d depth %d splitK %-5s : %d" % (k[0], k[1], k[2], k[3], used[k]))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
