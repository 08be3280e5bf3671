"""einsum on PyTorch-ROCm tensors through the engine — the counterpart of the reference's
`cutensor.torch.einsum` (cuTENSOR/python/cutensor/torch/einsum.py:25-156, einsum.cc:75-137).

PyTorch only provides device memory and the current HIP stream; parsing, planning and execution are
the C++ helper `cutensor_amd::Einsum<>` (csrc/einsum/einsum.hpp) driven through its C entry points.
"""
import ctypes

import torch

from . import cutensor as ct

_TORCH2CT = {torch.float32: ct.R_32F, torch.float64: ct.R_64F, torch.float16: ct.R_16F, torch.bfloat16: ct.R_16BF,
             torch.complex64: ct.C_32F, torch.complex128: ct.C_64F}   # complex: contractions, and unary equations (reduce / permute)

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

    def __init__(self, equation, a_shape, b_shape, dtype, conj_a=False, conj_b=False):
        self.e = ct.lib.ctamdEinsumCreate(equation.encode(), ct.i64(list(a_shape)), len(a_shape),
                                          ct.i64(list(b_shape)), len(b_shape), _TORCH2CT[dtype])
        if not self.e or not ct.lib.ctamdEinsumIsInitialized(self.e):
            raise ValueError("cutensor einsum: '%s' not supported for shapes %s, %s" % (equation, tuple(a_shape), tuple(b_shape)))
        if conj_a or conj_b:
            ct.lib.ctamdEinsumSetConjugate(self.e, int(conj_a), int(conj_b))
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

    def replan_min_workspace(self):
        """The reference binding's fallback (cutensor/torch/einsum.cc:104-123): when the workspace of the first plan
        cannot be allocated, plan again with CUTENSOR_WORKSPACE_MIN (here: limit 0) and use what that plan needs."""
        req = ctypes.c_uint64(0)
        if not ct.lib.ctamdEinsumReplan(self.e, get_handle(), 0, ctypes.byref(req)):
            raise RuntimeError("cutensor: plan with less workspace failed.")
        self.required_workspace = req.value

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


def einsum(equation, a, b=None, conj_a=False, conj_b=False):
    """out = einsum(equation, a[, b]) on the GPU holding `a`.  Inputs must be contiguous.  conj_a / conj_b conjugate an
    operand inside the contraction (python/cutensor/torch/einsum.py:50-61 uses them for complex gradients)."""
    if not a.is_cuda:
        raise RuntimeError("cutensor einsum runs on the GPU only (there is no CPU path)")
    a = a.contiguous()
    b = b.contiguous() if b is not None else None
    key = (equation, tuple(a.shape), tuple(b.shape) if b is not None else (), a.dtype, bool(conj_a), bool(conj_b))
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = EinsumPlan(equation, a.shape, b.shape if b is not None else (), a.dtype, conj_a, conj_b)
    out = torch.empty(plan.output_shape, dtype=a.dtype, device=a.device)
    try:
        ws = _get_workspace(a.device, plan.required_workspace)
    except torch.cuda.OutOfMemoryError:
        # einsum.cc:104-123: the workspace the plan asked for cannot be allocated -> plan again with the minimum
        plan.replan_min_workspace()
        try:
            ws = _get_workspace(a.device, plan.required_workspace)
        except torch.cuda.OutOfMemoryError:
            raise RuntimeError("cutensor: error allocating workspace")
    plan.execute(a, b, out, ws)
    return out


def _alloc_workspace(nbytes, device):
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _get_workspace(device, nbytes):
    """One growing workspace buffer per device, exactly as large as the largest plan so far asked for (the plan reports
    CUTENSOR_PLAN_REQUIRED_WORKSPACE; python/einsum.h:365-392)."""
    if not nbytes:
        return None
    ws = _workspace.get(device)
    if ws is None or ws.numel() < nbytes:
        _workspace.pop(device, None)
        ws = _workspace[device] = _alloc_workspace(nbytes, device)
    return ws


# ---------------------------------------------------------------------------------------------------------------
# Autograd, module and N-ary front ends — the counterparts of cuTENSOR/python/cutensor/torch/einsum.py:25-156
# (EinsumFunction.forward/backward :27-69, Einsum module :98-119, EinsumGeneral :139-172) and
# cutensor/common.py:18-28 (normalize_subscript).  Every contraction, reduction and permutation below runs
# through `einsum()` above, i.e. through the C ABI on the GPU; nothing falls back to torch.einsum.
# ---------------------------------------------------------------------------------------------------------------
def normalize_subscript(subscript):
    """'ik,kj' -> ('ik,kj->ij', True): implicit output = sorted modes that appear once (common.py:18-28)."""
    if "->" in subscript:
        lhs, rhs = subscript.split("->")
    else:
        lhs = subscript
        rhs = "".join(sorted(s for s in set(subscript) if s != "," and subscript.count(s) == 1))
    if "..." in lhs:
        raise RuntimeError("Elipsis is currently unsupported")
    return lhs + "->" + rhs, "," in lhs


class EinsumFunction(torch.autograd.Function):
    """out = einsum(equation, a[, b]) with gradients computed by further einsums on the engine
    (einsum.py:27-69: d_a = einsum(C,B->A)(grad, b), d_b = einsum(A,C->B)(a, grad); unary: grad permuted back)."""

    @staticmethod
    def forward(ctx, equation, input_0, input_1=None):
        equation, is_binary = normalize_subscript(equation)
        if is_binary and input_1 is None:
            raise RuntimeError("The subscript indicates two inputs, but only one was passed")
        if not is_binary and input_1 is not None:
            raise RuntimeError("The subscript indicates one input, but two were passed")
        out = einsum(equation, input_0.detach(), input_1.detach() if input_1 is not None else None)
        if is_binary:
            ctx.save_for_backward(input_0, input_1)
        else:
            ctx.in_shape = tuple(input_0.shape)
        ctx.equation, ctx.is_binary = equation, is_binary
        return out

    @staticmethod
    def backward(ctx, grad_output):
        lhs, mode_c = ctx.equation.split("->")
        grad_output = grad_output.contiguous()
        if ctx.is_binary:
            a, b = ctx.saved_tensors
            mode_a, mode_b = lhs.split(",")
            cj = torch.is_complex(a) or torch.is_complex(b)      # einsum.py:53-61: conjugate the saved operand
            d_a = _grad_einsum(mode_c, grad_output, mode_b, b.detach(), mode_a, a.shape, False, cj) if ctx.needs_input_grad[1] else None
            d_b = _grad_einsum(mode_a, a.detach(), mode_c, grad_output, mode_b, b.shape, cj, False) if ctx.needs_input_grad[2] else None
            return None, d_a, d_b
        d_in = _grad_einsum(mode_c, grad_output, None, None, lhs, ctx.in_shape) if ctx.needs_input_grad[1] else None
        return None, d_in, None


def _grad_einsum(m0, t0, m1, t1, m_out, out_shape, conj0=False, conj1=False):
    """einsum(m0[,m1] -> m_out); modes of m_out that neither operand carries (they were summed away in the
    forward pass) are broadcast afterwards — a stride-0 view, no copy."""
    present = [c for c in m_out if c in m0 or (m1 is not None and c in m1)]
    eq = m0 + ("," + m1 if m1 is not None else "") + "->" + "".join(present)
    g = einsum(eq, t0, t1, conj0, conj1)
    if len(present) == len(m_out):
        return g
    view = [out_shape[i] if c in present else 1 for i, c in enumerate(m_out)]
    return g.reshape(view).expand(out_shape)


class Einsum(torch.nn.Module):
    """torch.nn.Module wrapper (einsum.py:98-119)."""

    def __init__(self, equation):
        super().__init__()
        self.equation = equation

    def forward(self, input_0, input_1=None):
        return EinsumFunction.apply(self.equation, input_0, input_1)


def _compute_target_tensor(in0, in1, target, rest):
    """Modes the pairwise intermediate must keep: those still needed by the remaining operands or the output,
    output modes in output order (einsum.py:122-136)."""
    rest = "".join(rest) + target
    result = []
    for m in in0 + in1:
        if m in rest and m not in result:
            result.append(m)
    keep = sorted((m for m in result if m in target), key=target.index)
    it = iter(keep)
    return "".join(next(it) if m in target else m for m in result)


def EinsumGeneral(equation, *tensors, **kwargs):
    """N-ary einsum as a sequence of pairwise EinsumFunction calls along numpy's contraction path
    (einsum.py:139-172); differentiable end to end."""
    import numpy as np
    tensors = list(tensors)
    equation, _ = normalize_subscript(equation)
    lhs, target = equation.split("->")
    eqs = lhs.split(",")
    if len(eqs) != len(tensors):
        raise RuntimeError("The subscript indicates %d inputs, but %d were passed" % (len(eqs), len(tensors)))
    if len(tensors) == 1:
        return EinsumFunction.apply(eqs[0] + "->" + target, tensors[0])
    path = np.einsum_path(equation, *[np.broadcast_to(np.nan, t.shape) for t in tensors], **kwargs)[0][1:]
    result = None
    for step in path:
        if len(step) == 1:
            result = EinsumFunction.apply(eqs[step[0]] + "->" + target, tensors[step[0]])
            continue
        i, j = sorted(step)
        in0, in1 = tensors[i], tensors[j]
        e0, e1 = eqs[i], eqs[j]
        for k in (j, i):
            tensors.pop(k)
            eqs.pop(k)
        tgt = target if not tensors else _compute_target_tensor(e0, e1, target, eqs)
        result = EinsumFunction.apply(e0 + "," + e1 + "->" + tgt, in0, in1)
        tensors.append(result)
        eqs.append(tgt)
    return result
