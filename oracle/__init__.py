"""CPU oracle — TEST INFRASTRUCTURE ONLY.

ctypes/numpy front end of oracle/oracle.c (strided N-mode contraction / reduction / permutation with
fp64 accumulation, plus the einsum.cu equation front end).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; the product package never does.

Tensors are numpy arrays whose axes are labelled by a mode string/list; extents and element strides
are taken from the array itself, so any memory layout (C, F, sliced) is described exactly the way a
cuTENSOR tensor descriptor would describe it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-o", _SO, src, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _modes(m):
    return [ord(c) if isinstance(c, str) else int(c) for c in m]


def _desc(arr, modes):
    modes = _modes(modes)
    if arr.ndim != len(modes):
        raise ValueError("mode list %r does not match array rank %d" % (modes, arr.ndim))
    n = arr.ndim
    ext = (ctypes.c_int64 * max(n, 1))(*arr.shape)
    strides = [s // arr.itemsize for s in arr.strides]
    if any(s < 0 for s in strides):
        raise ValueError("negative strides are not supported")
    st = (ctypes.c_int64 * max(n, 1))(*strides)
    md = (ctypes.c_int32 * max(n, 1))(*modes)
    return n, md, ext, st


def _ptr(arr):
    return ctypes.c_void_p(arr.ctypes.data)


def contract(A, modesA, B, modesB, D, modesD, alpha=1.0, beta=0.0, C=None, acc64=True, conjA=False, conjB=False, h16=None):
    """D[modesD] = alpha * sum A[modesA]*B[modesB] + beta*C[modesD]; D is written in place.
    float32 / float64 / complex64 / complex128 arrays; h16 = "bf16" | "f16" with uint16 arrays of bit patterns (16-bit
    data, fp64 accumulation, one rounding of the result); conjA / conjB for complex data (CUTENSOR_OP_CONJ)."""
    if C is None:
        C = D
    if A.dtype != B.dtype or A.dtype != D.dtype or C.dtype != D.dtype:
        raise ValueError("dtype mismatch")
    if h16 is not None or np.iscomplexobj(A):
        if h16 is not None:
            if A.dtype != np.uint16:
                raise ValueError("16-bit tensors are passed as uint16 bit patterns")
            fn = {"bf16": lib().oracle_contract_bf16, "f16": lib().oracle_contract_f16}[h16]
        else:
            fn = lib().oracle_contract_c32 if A.dtype == np.complex64 else lib().oracle_contract_c64
        nA, mA, eA, sA = _desc(A, modesA)
        nB, mB, eB, sB = _desc(B, modesB)
        nC, mC, eC, sC = _desc(C, modesD)
        _, _, _, sD = _desc(D, modesD)
        fn.restype = ctypes.c_int
        if h16 is not None:
            rc = fn(nA, mA, eA, sA, _ptr(A), nB, mB, eB, sB, _ptr(B), nC, mC, eC, sC, _ptr(C), sD, _ptr(D),
                    ctypes.c_double(alpha), ctypes.c_double(beta))
        else:
            al, be = complex(alpha), complex(beta)
            rc = fn(nA, mA, eA, sA, _ptr(A), int(conjA), nB, mB, eB, sB, _ptr(B), int(conjB), nC, mC, eC, sC, _ptr(C), sD, _ptr(D),
                    ctypes.c_double(al.real), ctypes.c_double(al.imag), ctypes.c_double(be.real), ctypes.c_double(be.imag))
        if rc != 0:
            raise RuntimeError("oracle_contract failed: %d" % rc)
        return D
    if A.dtype == np.float32:
        fn = lib().oracle_contract_f32 if acc64 else lib().oracle_contract_f32_naive
    elif A.dtype == np.float64:
        fn = lib().oracle_contract_f64
    else:
        raise ValueError("oracle supports float32/float64")
    nA, mA, eA, sA = _desc(A, modesA)
    nB, mB, eB, sB = _desc(B, modesB)
    nC, mC, eC, sC = _desc(C, modesD)
    _, _, eD, sD = _desc(D, modesD)
    if tuple(C.shape) != tuple(D.shape):
        raise ValueError("C and D shapes differ")
    fn.restype = ctypes.c_int
    rc = fn(nA, mA, eA, sA, _ptr(A), nB, mB, eB, sB, _ptr(B), nC, mC, eC, sC, _ptr(C), sD, _ptr(D),
            ctypes.c_double(alpha), ctypes.c_double(beta))
    if rc != 0:
        raise RuntimeError("oracle_contract failed: %d" % rc)
    return D


OP_ADD, OP_MUL, OP_MAX, OP_MIN = 3, 5, 6, 7


def reduce(A, modesA, D, modesD, alpha=1.0, beta=0.0, C=None, op=OP_ADD, conjA=False, conjC=False):
    """D[modesD] = alpha * reduce_op over the modes of A missing from D + beta * C.  Complex tensors (complex64 / complex128:
    what python/einsum.h:326-343 hands to cutensorCreateReduction for a unary equation) take complex alpha / beta and the
    conjugation flags; op is ADD or MUL there."""
    if C is None:
        C = D
    nA, mA, eA, sA = _desc(A, modesA)
    nC, mC, eC, sC = _desc(C, modesD)
    _, _, _, sD = _desc(D, modesD)
    if np.issubdtype(A.dtype, np.complexfloating):
        fn = {np.dtype(np.complex64): lib().oracle_reduce_c32, np.dtype(np.complex128): lib().oracle_reduce_c64}[A.dtype]
        fn.restype = ctypes.c_int
        a, b = complex(alpha), complex(beta)
        rc = fn(nA, mA, eA, sA, _ptr(A), int(bool(conjA)), nC, mC, eC, sC, _ptr(C), int(bool(conjC)), sD, _ptr(D),
                ctypes.c_double(a.real), ctypes.c_double(a.imag), ctypes.c_double(b.real), ctypes.c_double(b.imag), int(op))
    else:
        fn = {np.dtype(np.float32): lib().oracle_reduce_f32, np.dtype(np.float64): lib().oracle_reduce_f64}[A.dtype]
        fn.restype = ctypes.c_int
        rc = fn(nA, mA, eA, sA, _ptr(A), nC, mC, eC, sC, _ptr(C), sD, _ptr(D), ctypes.c_double(alpha),
                ctypes.c_double(beta), int(op))
    if rc != 0:
        raise RuntimeError("oracle_reduce failed: %d" % rc)
    return D


def permute(A, modesA, B, modesB, alpha=1.0, C=None, gamma=0.0, conjA=False):
    nA, mA, eA, sA = _desc(A, modesA)
    nB, mB, eB, sB = _desc(B, modesB)
    if C is not None:
        _, _, _, sC = _desc(C, modesB)
        cp = _ptr(C)
    else:
        sC, cp = None, None
    if np.issubdtype(A.dtype, np.complexfloating):
        fn = {np.dtype(np.complex64): lib().oracle_permute_c32, np.dtype(np.complex128): lib().oracle_permute_c64}[A.dtype]
        fn.restype = ctypes.c_int
        a, g = complex(alpha), complex(gamma)
        rc = fn(nA, mA, eA, sA, _ptr(A), int(bool(conjA)), nB, mB, eB, sB, _ptr(B), ctypes.c_double(a.real), ctypes.c_double(a.imag),
                cp, sC, ctypes.c_double(g.real), ctypes.c_double(g.imag))
    else:
        fn = {np.dtype(np.float32): lib().oracle_permute_f32, np.dtype(np.float64): lib().oracle_permute_f64}[A.dtype]
        fn.restype = ctypes.c_int
        rc = fn(nA, mA, eA, sA, _ptr(A), nB, mB, eB, sB, _ptr(B), ctypes.c_double(alpha), cp, sC,
                ctypes.c_double(gamma))
    if rc != 0:
        raise RuntimeError("oracle_permute failed: %d" % rc)
    return B


def einsum_parse(equation, shapeA, shapeB=(), max_modes=40):
    """Restatement of Einsum::Einsum (einsum.cu:63-223).  Returns None when "not supported", else a
    dict with the REVERSED (cuTENSOR column-major order) mode/extent lists and the row-major output shape."""
    nA, nB = len(shapeA), len(shapeB)
    sa = (ctypes.c_int64 * max(nA, 1))(*shapeA)
    sb = (ctypes.c_int64 * max(nB, 1))(*shapeB)
    mA = (ctypes.c_int32 * 66)(); eA = (ctypes.c_int64 * 66)()
    mB = (ctypes.c_int32 * 66)(); eB = (ctypes.c_int64 * 66)()
    mC = (ctypes.c_int32 * 66)(); eC = (ctypes.c_int64 * 66)()
    nC = ctypes.c_int(0)
    fn = lib().oracle_einsum_parse
    fn.restype = ctypes.c_int
    ok = fn(equation.encode(), nA, sa, nB, sb, int(max_modes), mA, eA, mB, eB, ctypes.byref(nC), mC, eC)
    if not ok:
        return None
    n = nC.value
    return {
        "modesA": [chr(mA[i]) for i in range(nA)], "extentA": [eA[i] for i in range(nA)],
        "modesB": [chr(mB[i]) for i in range(nB)], "extentB": [eB[i] for i in range(nB)],
        "modesC": [chr(mC[i]) for i in range(n)], "extentC": [eC[i] for i in range(n)],
        "output_shape": [eC[n - 1 - i] for i in range(n)],
    }


def einsum(equation, a, b=None, h16=None, max_modes=40):
    """Framework-level einsum through the oracle (row-major numpy arrays in, row-major out).  float32 / float64 /
    complex64 / complex128 arrays, or uint16 bit patterns with h16 = "bf16" | "f16" (two-operand equations)."""
    p = einsum_parse(equation, a.shape, b.shape if b is not None else (), max_modes=max_modes)
    if p is None:
        raise ValueError("not supported: %s" % equation)
    out = np.zeros(p["output_shape"], dtype=a.dtype)
    # numpy axis i of a row-major array carries mode modes[::-1][i]
    ma, mc = p["modesA"][::-1], p["modesC"][::-1]
    if b is not None:
        contract(a, ma, b, p["modesB"][::-1], out, mc, h16=h16)
    else:
        reduce(a, ma, out, mc)
    return out


def to_bits(x, kind):
    """float array -> uint16 bit patterns of the 16-bit type (round to nearest even), through the oracle's own conversion."""
    f = lib().oracle_double_to_bf16 if kind == "bf16" else lib().oracle_double_to_f16
    f.restype = ctypes.c_uint16
    f.argtypes = [ctypes.c_double]
    flat = np.asarray(x, dtype=np.float64).ravel(order="K")
    out = np.fromiter((f(float(v)) for v in flat), dtype=np.uint16, count=flat.size)
    return out.reshape(np.shape(x), order="F" if np.isfortran(np.asarray(x)) else "C")


def from_bits(u, kind):
    """uint16 bit patterns -> float64 values (exact)."""
    u = np.asarray(u, dtype=np.uint16)
    if kind == "f16":
        return u.view(np.float16).astype(np.float64)
    return (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
