"""einsum on PyTorch-ROCm tensors through the engine — the counterpart of the reference's
`cutensor.torch.einsum` (cuTENSOR/python/cutensor/torch/einsum.py:25-156, einsum.cc:75-137).

PyTorch only provides device memory and the current HIP stream; parsing, planning and execution are
the C++ helper `cutensor_amd::Einsum<>` (csrc/einsum/einsum.hpp) driven through its C entry points.
"""
import ctypes

import torch

from . import cutensor as ct

_TORCH2CT = {torch.float32: ct.R_32F, torch.float64: ct.R_64F, torch.float16: ct.R_16F, torch.bfloat16: ct.R_16BF}

_handle = None


def get_handle():
    """Process-wide handle, as GetCuTensorHandle() in cuTENSOR/python/einsum.h:484-502."""
    global _handle
    if _handle is None:
        h = ctypes.c_void_p()
        ct.check(ct.cutensorCreate(ctypes.byref(h)))
        ct.check(ct.cutensorHandleResizePlanCache(h, 1024))
        _handle = h
    return _handle


class EinsumPlan:
    """Parsed + planned equation for fixed shapes/dtype (plan()/execute() split of python/einsum.h)."""

    def __init__(self, equation, a_shape, b_shape, dtype):
        self.e = ct.lib.ctamdEinsumCreate(equation.encode(), ct.i64(list(a_shape)), len(a_shape),
                                          ct.i64(list(b_shape)), len(b_shape), _TORCH2CT[dtype])
        if not self.e or not ct.lib.ctamdEinsumIsInitialized(self.e):
            raise ValueError("cutensor einsum: '%s' not supported for shapes %s, %s" % (equation, tuple(a_shape), tuple(b_shape)))
        out = (ctypes.c_int64 * 64)()
        n = ct.lib.ctamdEinsumOutputShape(self.e, out, 64)
        self.output_shape = [out[i] for i in range(n)]
        req = ctypes.c_uint64(0)
        if not ct.lib.ctamdEinsumPlan(self.e, get_handle(), 1 << 30, ctypes.byref(req)):
            raise RuntimeError("cutensor einsum: planning failed for '%s'" % equation)
        self.required_workspace = req.value
        self.dtype = dtype

    def describe(self):
        return ct.describe_plan(ct.lib.ctamdEinsumRawPlan(self.e))

    def execute(self, a, b, out, workspace):
        stream = torch.cuda.current_stream().cuda_stream
        ok = ct.lib.ctamdEinsumExecute(self.e, get_handle(), a.data_ptr(), b.data_ptr() if b is not None else None,
                                       out.data_ptr(), workspace.data_ptr() if workspace is not None else None, stream)
        if not ok:
            raise RuntimeError("cutensor einsum: execution failed")

    def __del__(self):
        try:
            if self.e:
                ct.lib.ctamdEinsumDestroy(self.e)
        except Exception:
            pass


_plans = {}
_workspace = {}


def einsum(equation, a, b=None):
    """out = einsum(equation, a[, b]) on the GPU holding `a`.  Inputs must be contiguous."""
    if not a.is_cuda:
        raise RuntimeError("cutensor einsum runs on the GPU only (there is no CPU path)")
    a = a.contiguous()
    b = b.contiguous() if b is not None else None
    key = (equation, tuple(a.shape), tuple(b.shape) if b is not None else (), a.dtype)
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = EinsumPlan(equation, a.shape, b.shape if b is not None else (), a.dtype)
    out = torch.empty(plan.output_shape, dtype=a.dtype, device=a.device)
    ws = None
    if plan.required_workspace:
        # the helper passes its fixed 1 GiB worksize to the ABI (einsum.cu:380); provide that much once
        ws = _workspace.get(a.device)
        if ws is None:
            ws = _workspace[a.device] = torch.empty(1 << 30, dtype=torch.uint8, device=a.device)
    plan.execute(a, b, out, ws)
    return out
