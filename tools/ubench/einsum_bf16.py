"""Headline einsum shape 'abcd,dcbe->ae' (a=e=96, b=c=d=64) with bf16 data: split-K path of the 16-bit GETT kernel."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from cudalibrarysamples_amd import cutensor as ct, ops
ext = dict(a=96, b=64, c=64, d=64, e=96)
h = ops.Handle()
plan = ops.contraction_plan(h, [ext[c] for c in "dcba"], "dcba", [ext[c] for c in "ebcd"], "ebcd", [ext[c] for c in "ea"], "ea",
                            dtype=ct.R_16BF, workspace_limit=1 << 30)
print(plan.describe())
A = (torch.rand(96 * 64 ** 3, device="cuda") * 2 - 1).to(torch.bfloat16)
B = (torch.rand(96 * 64 ** 3, device="cuda") * 2 - 1).to(torch.bfloat16)
C = torch.zeros(96 * 96, device="cuda", dtype=torch.bfloat16)
ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(1000):
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), plan.required_workspace, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(2000):
    plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, C.data_ptr(), C.data_ptr(), ws.data_ptr(), plan.required_workspace, s)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 2000 * 1e3
print("bf16 einsum abcd,dcbe->ae: %.1f us per call = %.1f TFLOP/s, %.2f TB/s of operand bytes" % (us, 4.8318e9 / us / 1e6, 100.7e6 / us / 1e6))
