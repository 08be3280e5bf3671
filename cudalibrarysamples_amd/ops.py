"""Convenience layer over the ABI for device buffers given as raw pointers.

Everything here is a straight replay of the reference samples' call sequence
(cuTENSOR/contraction.cu:122-265, reduction.cu:96-222, elementwise_permute.cu:100-200): create
descriptors -> operation descriptor -> plan preference -> workspace estimate -> plan -> execute.
PyTorch appears only as the owner of device memory and streams in the callers.
"""
import ctypes

from . import cutensor as ct

_DTYPE_COMPUTE = {ct.R_32F: "32F", ct.R_64F: "64F", ct.R_16F: "16F", ct.R_16BF: "16BF", ct.C_32F: "32F", ct.C_64F: "64F"}


class Handle:
    def __init__(self, plan_cache=0):
        self.h = ctypes.c_void_p()
        ct.check(ct.cutensorCreate(ctypes.byref(self.h)))
        if plan_cache:
            ct.check(ct.cutensorHandleResizePlanCache(self.h, plan_cache))

    def close(self):
        if self.h:
            ct.cutensorDestroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tensor_descriptor(handle, extent, stride=None, dtype=ct.R_32F, alignment=128):
    d = ctypes.c_void_p()
    ct.check(ct.cutensorCreateTensorDescriptor(handle.h, ctypes.byref(d), len(extent), ct.i64(extent),
                                               ct.i64(stride) if stride is not None else None, dtype, alignment))
    return d


class Plan:
    """An operation descriptor + plan pair with its scalar type and workspace requirement."""

    def __init__(self, handle, op, kind, dtype, algo=ct.ALGO_DEFAULT, kernel_rank=0, workspace_limit=None,
                 workspace_pref=ct.WORKSPACE_DEFAULT, autotune=None, cache_mode=None, incremental_count=None, operands_streamed=None):
        self.handle, self.kind, self.dtype = handle, kind, dtype
        self.op = op
        pref = ctypes.c_void_p()
        ct.check(ct.cutensorCreatePlanPreference(handle.h, ctypes.byref(pref), algo, ct.JIT_MODE_NONE))
        for attr, val in ((ct.PLAN_PREFERENCE_KERNEL_RANK, kernel_rank or None), (ct.PLAN_PREFERENCE_AUTOTUNE_MODE, autotune),
                          (ct.PLAN_PREFERENCE_CACHE_MODE, cache_mode), (ct.PLAN_PREFERENCE_INCREMENTAL_COUNT, incremental_count),
                          (ct.AMD_PLAN_PREFERENCE_OPERANDS_STREAMED, (1 if operands_streamed else None))):
            if val is not None:   # contraction_plan_cache.cu:215-237
                v = ctypes.c_int32(val)
                ct.check(ct.cutensorPlanPreferenceSetAttribute(handle.h, pref, attr, ctypes.byref(v), 4))
        est = ctypes.c_uint64(0)
        ct.check(ct.cutensorEstimateWorkspaceSize(handle.h, op, pref, workspace_pref, ctypes.byref(est)))
        self.workspace_estimate = est.value
        limit = est.value if workspace_limit is None else workspace_limit
        self.plan = ctypes.c_void_p()
        st = ct.cutensorCreatePlan(handle.h, ctypes.byref(self.plan), op, pref, limit)
        ct.cutensorDestroyPlanPreference(pref)
        ct.check(st)
        req = ctypes.c_uint64(0)
        ct.check(ct.cutensorPlanGetAttribute(handle.h, self.plan, ct.PLAN_REQUIRED_WORKSPACE, ctypes.byref(req), 8))
        self.required_workspace = req.value
        st_type = ctypes.c_int(0)
        ct.check(ct.cutensorOperationDescriptorGetAttribute(handle.h, op, ct.OPERATION_DESCRIPTOR_SCALAR_TYPE,
                                                            ctypes.byref(st_type), 4))
        self.scalar_type = st_type.value

    def scalar(self, x):
        if self.scalar_type == ct.C_32F:
            return (ctypes.c_float * 2)(complex(x).real, complex(x).imag)
        if self.scalar_type == ct.C_64F:
            return (ctypes.c_double * 2)(complex(x).real, complex(x).imag)
        return ctypes.c_double(x) if self.scalar_type == ct.R_64F else ctypes.c_float(x)

    def describe(self):
        return ct.describe_plan(self.plan)

    def destroy(self):
        if self.plan:
            ct.cutensorDestroyPlan(self.plan)
            self.plan = ctypes.c_void_p()
        if self.op:
            ct.cutensorDestroyOperationDescriptor(self.op)
            self.op = ctypes.c_void_p()

    # ---- execution (raw device pointers as ints) ------------------------------------------------
    def contract(self, alpha, A, B, beta, C, D, workspace=0, workspace_size=0, stream=0):
        a, b = self.scalar(alpha), self.scalar(beta)
        ct.check(ct.cutensorContract(self.handle.h, self.plan, ctypes.byref(a), A, B, ctypes.byref(b), C, D,
                                     workspace or None, workspace_size, stream or None))

    def reduce(self, alpha, A, beta, C, D, workspace=0, workspace_size=0, stream=0):
        a, b = self.scalar(alpha), self.scalar(beta)
        ct.check(ct.cutensorReduce(self.handle.h, self.plan, ctypes.byref(a), A, ctypes.byref(b), C, D,
                                   workspace or None, workspace_size, stream or None))

    def permute(self, alpha, A, B, stream=0):
        a = self.scalar(alpha)
        ct.check(ct.cutensorPermute(self.handle.h, self.plan, ctypes.byref(a), A, B, stream or None))

    def contract_trinary(self, alpha, A, B, C, beta, D, E, workspace=0, workspace_size=0, stream=0):
        a, b = self.scalar(alpha), self.scalar(beta)
        ct.check(ct.cutensorContractTrinary(self.handle.h, self.plan, ctypes.byref(a), A, B, C, ctypes.byref(b), D, E,
                                            workspace or None, workspace_size, stream or None))

    def trinary(self, alpha, A, beta, B, gamma, C, D, stream=0):
        a, b, g = self.scalar(alpha), self.scalar(beta), self.scalar(gamma)
        ct.check(ct.cutensorElementwiseTrinaryExecute(self.handle.h, self.plan, ctypes.byref(a), A, ctypes.byref(b), B,
                                                      ctypes.byref(g), C, D, stream or None))

    def binary(self, alpha, A, gamma, C, D, stream=0):
        a, g = self.scalar(alpha), self.scalar(gamma)
        ct.check(ct.cutensorElementwiseBinaryExecute(self.handle.h, self.plan, ctypes.byref(a), A, ctypes.byref(g),
                                                     C, D, stream or None))


_OPS = {"ADD": ct.OP_ADD, "MUL": ct.OP_MUL, "MAX": ct.OP_MAX, "MIN": ct.OP_MIN}


def _desc3(handle, specs, dtype, alignment):
    return [tensor_descriptor(handle, e, s, dtype, alignment) for (e, s) in specs]


def contraction_plan(handle, extA, modesA, extB, modesB, extC, modesC, dtype=ct.R_32F, strideA=None,
                     strideB=None, strideC=None, strideD=None, compute=None, alignment=128, opA=ct.OP_IDENTITY,
                     opB=ct.OP_IDENTITY, opC=ct.OP_IDENTITY, **plan_kw):
    dA, dB, dC = _desc3(handle, [(extA, strideA), (extB, strideB), (extC, strideC)], dtype, alignment)
    dD = tensor_descriptor(handle, extC, strideD, dtype, alignment) if strideD is not None else dC
    op = ctypes.c_void_p()
    st = ct.cutensorCreateContraction(handle.h, ctypes.byref(op), dA, ct.i32(modesA), opA, dB, ct.i32(modesB),
                                      opB, dC, ct.i32(modesC), opC, dD, ct.i32(modesC),
                                      ct.compute_desc(compute or _DTYPE_COMPUTE[dtype]))
    for d in {id(x): x for x in (dA, dB, dC, dD)}.values():
        ct.cutensorDestroyTensorDescriptor(d)
    ct.check(st)
    return Plan(handle, op, "contraction", dtype, **plan_kw)


def reduction_plan(handle, extA, modesA, extC, modesC, dtype=ct.R_32F, strideA=None, strideC=None,
                   op_reduce=ct.OP_ADD, compute=None, alignment=128, opA=ct.OP_IDENTITY, opC=ct.OP_IDENTITY, **plan_kw):
    dA, dC = _desc3(handle, [(extA, strideA), (extC, strideC)], dtype, alignment)
    op = ctypes.c_void_p()
    st = ct.cutensorCreateReduction(handle.h, ctypes.byref(op), dA, ct.i32(modesA), opA, dC, ct.i32(modesC),
                                    opC, dC, ct.i32(modesC), op_reduce,
                                    ct.compute_desc(compute or _DTYPE_COMPUTE[dtype]))
    ct.cutensorDestroyTensorDescriptor(dA)
    ct.cutensorDestroyTensorDescriptor(dC)
    ct.check(st)
    return Plan(handle, op, "reduction", dtype, **plan_kw)


def permutation_plan(handle, extA, modesA, extB, modesB, dtype=ct.R_32F, strideA=None, strideB=None,
                     compute=None, alignment=128, padding=None, opA=ct.OP_IDENTITY, **plan_kw):
    """padding = (left[], right[], value): CUTENSOR_OPERATION_DESCRIPTOR_PADDING_* per output mode
    (elementwise_permute_padding.cu:178-195); the output buffer then holds extB + left + right per mode."""
    dA, dB = _desc3(handle, [(extA, strideA), (extB, strideB)], dtype, alignment)
    op = ctypes.c_void_p()
    st = ct.cutensorCreatePermutation(handle.h, ctypes.byref(op), dA, ct.i32(modesA), opA, dB, ct.i32(modesB),
                                      ct.compute_desc(compute or _DTYPE_COMPUTE[dtype]))
    ct.cutensorDestroyTensorDescriptor(dA)
    ct.cutensorDestroyTensorDescriptor(dB)
    ct.check(st)
    if padding is not None:
        left, right, value = padding
        n = len(extB)
        l = (ctypes.c_int32 * n)(*left)
        r = (ctypes.c_int32 * n)(*right)
        ct.check(ct.cutensorOperationDescriptorSetAttribute(handle.h, op, 4, l, 4 * n))    # PADDING_LEFT
        ct.check(ct.cutensorOperationDescriptorSetAttribute(handle.h, op, 5, r, 4 * n))    # PADDING_RIGHT
        v = ctypes.c_double(value) if dtype == ct.R_64F else ctypes.c_float(value)
        ct.check(ct.cutensorOperationDescriptorSetAttribute(handle.h, op, 6, ctypes.byref(v), ctypes.sizeof(v)))   # PADDING_VALUE
    plan_kw.setdefault("workspace_limit", 0)   # elementwise_permute.cu:183-187
    return Plan(handle, op, "permutation", dtype, **plan_kw)


def binary_plan(handle, extA, modesA, extC, modesC, op="ADD", dtype=ct.R_32F, compute=None, alignment=128, opA=ct.OP_IDENTITY,
                opC=ct.OP_IDENTITY, **plan_kw):
    """D = op(alpha * perm(A), gamma * C) — cutensorCreateElementwiseBinary (elementwise_binary.cu:149-153)."""
    dA, dC = _desc3(handle, [(extA, None), (extC, None)], dtype, alignment)
    opd = ctypes.c_void_p()
    st = ct.cutensorCreateElementwiseBinary(handle.h, ctypes.byref(opd), dA, ct.i32(modesA), opA, dC, ct.i32(modesC),
                                            opC, dC, ct.i32(modesC), _OPS[op],
                                            ct.compute_desc(compute or _DTYPE_COMPUTE[dtype]))
    ct.cutensorDestroyTensorDescriptor(dA)
    ct.cutensorDestroyTensorDescriptor(dC)
    ct.check(st)
    plan_kw.setdefault("workspace_limit", 0)
    return Plan(handle, opd, "binary", dtype, **plan_kw)


def trinary_plan(handle, extA, modesA, extB, modesB, extC, modesC, extD, modesD, opAB="ADD", opABC="ADD", dtype=ct.R_32F,
                 compute=None, alignment=128, **plan_kw):
    """D = opABC(opAB(alpha * perm(A), beta * perm(B)), gamma * perm(C)) — cutensorCreateElementwiseTrinary
    (elementwise_trinary.cu:174-182)."""
    dA, dB, dC, dD = _desc3(handle, [(extA, None), (extB, None), (extC, None), (extD, None)], dtype, alignment)
    opd = ctypes.c_void_p()
    st = ct.cutensorCreateElementwiseTrinary(handle.h, ctypes.byref(opd), dA, ct.i32(modesA), ct.OP_IDENTITY, dB, ct.i32(modesB),
                                             ct.OP_IDENTITY, dC, ct.i32(modesC), ct.OP_IDENTITY, dD, ct.i32(modesD),
                                             _OPS[opAB], _OPS[opABC], ct.compute_desc(compute or _DTYPE_COMPUTE[dtype]))
    for d in (dA, dB, dC, dD):
        ct.cutensorDestroyTensorDescriptor(d)
    ct.check(st)
    plan_kw.setdefault("workspace_limit", 0)
    return Plan(handle, opd, "trinary", dtype, **plan_kw)


def contraction_trinary_plan(handle, extA, modesA, extB, modesB, extC, modesC, extD, modesD, dtype=ct.R_32F, compute=None,
                             alignment=128, **plan_kw):
    """E = alpha * A * B * C + beta * D — cutensorCreateContractionTrinary (contraction_trinary.cu:191-198); E shares
    D's descriptor as in the sample."""
    dA, dB, dC, dD = _desc3(handle, [(extA, None), (extB, None), (extC, None), (extD, None)], dtype, alignment)
    opd = ctypes.c_void_p()
    st = ct.cutensorCreateContractionTrinary(handle.h, ctypes.byref(opd), dA, ct.i32(modesA), ct.OP_IDENTITY, dB, ct.i32(modesB),
                                             ct.OP_IDENTITY, dC, ct.i32(modesC), ct.OP_IDENTITY, dD, ct.i32(modesD), ct.OP_IDENTITY,
                                             dD, ct.i32(modesD), ct.compute_desc(compute or _DTYPE_COMPUTE[dtype]))
    for d in (dA, dB, dC, dD):
        ct.cutensorDestroyTensorDescriptor(d)
    ct.check(st)
    return Plan(handle, opd, "contraction_trinary", dtype, **plan_kw)
