#!/usr/bin/env python3
"""Randomised parity sweep of cutensorMpContract over the in-process local world (ranks as threads on one GPU): random
contractions, extents (ragged blocks included), rank counts, per-tensor distributions, rank permutations, data types,
alpha / beta, and both exchange algorithms.  Uses the checker of tests/test_gpu_mp.py.  Not part of the test suite."""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def factorisations(n, k):
    """All ways to write n as an ordered product of k factors."""
    if k == 1:
        return [[n]]
    out = []
    for f in range(1, n + 1):
        if n % f == 0:
            out += [[f] + rest for rest in factorisations(n // f, k - 1)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=150)
    args = ap.parse_args()
    import torch
    from cudalibrarysamples_amd import cutensor as ct, cutensormp
    import test_gpu_mp as T
    mp = (cutensormp, ct, torch)
    rnd = random.Random(args.seed)
    fails, algos = 0, {}
    for case in range(args.cases):
        nM, nN, nK, nL = rnd.randint(1, 2), rnd.randint(1, 2), rnd.randint(1, 2), rnd.choice([0, 0, 1])
        labels = list("abcdefgh")
        rnd.shuffle(labels)
        M, N, K, L = ([labels.pop() for _ in range(x)] for x in (nM, nN, nK, nL))
        ext = {c: rnd.choice([1, 2, 3, 4, 6, 8, 13, 16, 24]) for c in M + N + K + L}
        mA, mB, mC = M + K + L, N + K + L, M + N + L
        for m in (mA, mB, mC):
            rnd.shuffle(m)
        nranks = rnd.choice([2, 2, 3, 4])
        dist, ranks = [], []
        for m in (mA, mB, mC):
            if rnd.random() < 0.2:
                dist.append({})                                  # replicated
                ranks.append(None)
                continue
            f = rnd.choice(factorisations(nranks, len(m)))
            dist.append({c: p for c, p in zip(m, f)})
            perm = list(range(nranks))
            if rnd.random() < 0.3:
                rnd.shuffle(perm)
            ranks.append(perm)
        if rnd.random() < 0.35:      # K-distributed family: A and B cut identically along one contracted mode
            k = rnd.choice(K)
            ext[k] = rnd.choice([nranks, 2 * nranks, 8 * nranks, 8 * nranks + 1])
            perm = list(range(nranks))
            rnd.shuffle(perm)
            dist[0], dist[1] = {k: nranks}, {k: nranks}
            ranks[0], ranks[1] = perm, list(perm)
        dtype = rnd.choice(["f32", "f32", "f64", "c64", "bf16"])
        alpha, beta = rnd.choice([1.0, 0.5, -1.25]), rnd.choice([0.0, 0.0, 0.5])
        algo = rnd.choice(["", "gather", "reduce"])
        if algo:
            os.environ["CUTENSORMP_AMD_ALGO"] = algo
        else:
            os.environ.pop("CUTENSORMP_AMD_ALGO", None)
        eq = "%s,%s->%s" % ("".join(mA), "".join(mB), "".join(mC))
        try:
            d = T.run_case(mp, eq, ext, tuple(dist), nranks, dtype, alpha, beta, ranks=ranks, seed=case)
            key = d[0]["algorithm"]
            algos[key] = algos.get(key, 0) + 1
        except Exception as e:   # noqa: BLE001
            fails += 1
            print("case %d FAILED: %s %s dist %s ranks %s nranks %d %s alpha %g beta %g algo '%s': %s" % (
                case, eq, ext, dist, ranks, nranks, dtype, alpha, beta, algo, str(e)[:400]))
    print("cases %d, failures %d, algorithms %s" % (args.cases, fails, algos))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
