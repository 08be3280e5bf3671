"""TTGT-over-OpenBLAS CPU baseline (TEST INFRASTRUCTURE / bench cpu_baseline only).

Transpose-Transpose-GEMM-Transpose: materialise A as an (M x K) and B as a (K x N) matrix, run one
OpenBLAS sgemm through numpy, reshape back.  This is the CPU baseline BASELINE.md section 4 asks to
time next to every GPU number; the transposes are part of the timed region.
"""
import time

import numpy as np


def ttgt_einsum_abcd_dcbe_ae(a, b):
    """'abcd,dcbe->ae' on row-major arrays a[a,b,c,d], b[d,c,b,e]."""
    na = a.shape[0]
    ne = b.shape[3]
    am = a.reshape(na, -1)                                   # (a, bcd) — already a matrix
    bm = np.ascontiguousarray(b.transpose(2, 1, 0, 3)).reshape(-1, ne)   # (bcd, e): explicit transpose
    return am @ bm


def time_ttgt(a, b, reps=3):
    best, best_gemm = 1e30, 1e30
    out = None
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        am = a.reshape(a.shape[0], -1)
        bm = np.ascontiguousarray(b.transpose(2, 1, 0, 3)).reshape(-1, b.shape[3])
        t1 = time.perf_counter()
        out = am @ bm
        t2 = time.perf_counter()
        best = min(best, t2 - t0)
        best_gemm = min(best_gemm, t2 - t1)
    return out, best, best_gemm
