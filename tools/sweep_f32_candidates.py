import sys, json, torch
sys.path.insert(0, '/root/repo')
from cudalibrarysamples_amd import cutensor as ct, ops
h = ops.Handle()
E = int(sys.argv[1]); mA, mB = sys.argv[2].split(",")
g = torch.Generator(device="cuda"); g.manual_seed(1)
A = torch.rand((E, E), generator=g, device="cuda"); B = torch.rand((E, E), generator=g, device="cuda"); D = torch.zeros((E, E), device="cuda")
for r in range(int(sys.argv[3])):
    try:
        plan = ops.contraction_plan(h, [E, E], mA, [E, E], mB, [E, E], "mn", dtype=ct.R_32F, workspace_limit=1 << 30, algo=r)
    except Exception as e:
        print("rank", r, "fail", e); break
    d = plan.describe()
    ws = torch.empty(max(plan.required_workspace, 16), dtype=torch.uint8, device="cuda")
    run = lambda: plan.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), plan.required_workspace)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps(dict(rank=r, kname=d["kname"], kernel=d["kernel"], bm=d["bm"], bn=d["bn"], pf=d["pf"], splitK=d["splitK"], model_us=d["model_us"], us=round(ms * 1e3, 1), tflops=round(2.0 * E ** 3 / ms / 1e9, 1))), flush=True)
    plan.destroy()
