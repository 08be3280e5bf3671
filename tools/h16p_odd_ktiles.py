import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
g = torch.Generator(device="cuda"); g.manual_seed(1)
for (M, N, K) in ((4352, 4352, 192), (8192, 8192, 1088), (8192, 8192, 1024), (8192, 8192, 448), (8192, 8192, 128)):
    A = (torch.rand((K, M), generator=g, device="cuda") * 2 - 1).bfloat16(); B = (torch.rand((N, K), generator=g, device="cuda") * 2 - 1).bfloat16()
    D = torch.empty((N, M), dtype=torch.bfloat16, device="cuda")
    p = ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=ct.R_16BF)
    fn = lambda: p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr())
    for _ in range(100): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    print(json.dumps(dict(M=M, N=N, K=K, forced=os.environ.get("CUTENSOR_AMD_H16_WAVES"), kname=p.describe()["kname"], us=round(best * 1e3, 2), tflops=round(2.0 * M * N * K / best / 1e9, 1))), flush=True)
