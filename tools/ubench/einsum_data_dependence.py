import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops
ext = dict(a=96, b=64, c=64, d=64, e=96)
h = ops.Handle()
plan = ops.contraction_plan(h, [ext[c] for c in "dcba"], "dcba", [ext[c] for c in "ebcd"], "ebcd", [ext[c] for c in "ea"], "ea", workspace_limit=1 << 30)
ws = torch.empty(plan.required_workspace, dtype=torch.uint8, device="cuda")
C = torch.zeros(96 * 96, device="cuda")
for name, fill in (("random", None), ("zeros", 0.0), ("ones", 1.0)):
    A = torch.rand(96 * 64 ** 3, device="cuda"); B = torch.rand(96 * 64 ** 3, device="cuda")
    if fill is not None:
        A.fill_(fill); B.fill_(fill)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), plan.required_workspace, s)
    torch.cuda.synchronize()
    ct.lib.ctamdProfileBegin(h.h)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), plan.required_workspace, s)
    e1.record(); torch.cuda.synchronize()
    m, mn = ctypes.c_float(0), ctypes.c_float(0)
    ct.lib.ctamdProfileEnd(h.h, ctypes.byref(m), ctypes.byref(mn))
    print(name, "step %.2f us  kernel %.2f us (min %.2f)" % (e0.elapsed_time(e1) * 5, m.value * 1e3, mn.value * 1e3))
