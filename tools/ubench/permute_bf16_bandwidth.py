import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
n = 2048
A = torch.rand((n, n, n // 2), device="cuda").to(torch.bfloat16)     # 4.3 G elements of 2 B = 8.6 GB
for mD in ("cba", "acb"):
    ext = dict(a=n // 2, b=n, c=n)
    eA, eD = [ext[c] for c in "abc"], [ext[c] for c in mD]
    D = torch.empty(A.numel(), device="cuda", dtype=torch.bfloat16)
    plan = ops.permutation_plan(h, eA, "abc", eD, mD, dtype=ct.R_16BF)
    for _ in range(5):
        plan.permute(1.0, A.data_ptr(), D.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.permute(1.0, A.data_ptr(), D.data_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("bf16 permute abc->%s: %.2f ms = %.2f TB/s (variant %d)" % (mD, ms, 2 * A.numel() * 2 / ms / 1e9, plan.describe()["variant"]))
