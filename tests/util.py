"""Shared helpers for the parity tests: seeded tensors in cuTENSOR (packed column-major) layout,
device round trips through torch, and the oracle comparison."""
import ctypes

import numpy as np


def make_tensor(extents, seed, dtype=np.float32, lo=0.0, hi=1.0):
    """Packed generalized column-major tensor (first mode fastest), U(lo, hi), fixed seed.
    The reference samples draw U(0,1) from a nondeterministically seeded mt19937 (utils.cuh:76-113);
    the tests pin the seed instead."""
    rng = np.random.default_rng(seed)
    n = int(np.prod(extents)) if len(extents) else 1
    flat = (rng.random(n, dtype=np.float64) * (hi - lo) + lo).astype(dtype)
    return np.reshape(flat, tuple(extents), order="F") if len(extents) else flat.reshape(())


def to_device(arr):
    import torch
    flat = np.ascontiguousarray(arr.ravel(order="K")) if arr.ndim else arr.reshape(1)
    # ravel(order='K') walks memory order: for an F-ordered packed array that is the raw buffer
    return torch.from_numpy(flat.copy()).cuda()


def from_device(t, like):
    out = t.cpu().numpy()
    return np.reshape(out, like.shape, order="F") if like.ndim else out.reshape(())


def rel_err(got, ref):
    got = got.astype(np.float64)
    ref = ref.astype(np.float64)
    denom = np.maximum(np.abs(ref), 1e-30)
    return float(np.max(np.abs(got - ref) / np.maximum(denom, np.max(np.abs(ref)) * 1e-6)))


def assert_close(got, ref, rtol, atol=0.0, what=""):
    got64 = got.astype(np.float64)
    ref64 = ref.astype(np.float64)
    err = np.abs(got64 - ref64)
    tol = atol + rtol * np.abs(ref64)
    bad = err > tol
    if bad.any():
        idx = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: %d/%d elements off; worst at %s: got %r ref %r (rtol %g atol %g)" % (
            what, int(bad.sum()), bad.size, idx, got64[idx], ref64[idx], rtol, atol))


def c_i64(v):
    return (ctypes.c_int64 * max(len(v), 1))(*v)


def golden_inputs(dtype_name, a_size, b_size):
    """Inputs of a reference test case as tests/golden/make_golden.py draws them (einsum_test.py:131-147 on the CPU generator):
    torch.manual_seed(0), randn in the case's dtype (16-bit types: fp32 randn rounded once).  Returns (a, b) torch tensors of the
    case's dtype.  The full-extent fixtures (tests/golden/full) store no inputs, only probes of these to detect generator drift."""
    import torch
    tdt = getattr(torch, dtype_name)
    torch.manual_seed(0)
    if tdt.is_complex:
        return torch.randn(*a_size, dtype=tdt), torch.randn(*b_size, dtype=tdt)
    return torch.randn(*a_size, dtype=torch.float32).to(tdt), torch.randn(*b_size, dtype=torch.float32).to(tdt)
