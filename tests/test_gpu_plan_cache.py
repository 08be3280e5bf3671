"""Plan cache + incremental autotuning through the C ABI, as cuTENSOR/contraction_plan_cache.cu drives it: cache mode
PEDANTIC, CUTENSOR_AUTOTUNE_MODE_INCREMENTAL with INCREMENTAL_COUNT = 4 (:215-237), count + 1 plan / contract rounds of
which the last hits the cache (:262-318), cutensorHandleWritePlanCacheToFile (:324-337), and a fresh handle that reads
the file back (:132-154) and plans the tuned kernel without measuring anything."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import cutensor as ct, ops
    return ct, ops, torch


def _choice(d):
    return (d["kernel"], d["splitK"])


@pytest.mark.parametrize("dtype,M,N,K,expect_trials", [("f32", 512, 384, 4096, 4), ("bf16", 512, 512, 1024, 4)])
def test_incremental_autotune_fills_the_cache_and_the_file_round_trips(env, tmp_path, dtype, M, N, K, expect_trials):
    ct, ops, torch = env
    cdt = ct.R_32F if dtype == "f32" else ct.R_16BF
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    A = (torch.rand((K, M), generator=g, device="cuda") * 2 - 1).to(tdt)     # modes "mk": m fastest
    B = (torch.rand((N, K), generator=g, device="cuda") * 2 - 1).to(tdt)     # modes "kn"
    D = torch.zeros((N, M), device="cuda", dtype=tdt)                        # modes "mn"
    ref = (B.double() @ A.double()).cpu().numpy()
    tol = dict(rtol=1e-4, atol=1e-3) if dtype == "f32" else dict(rtol=2e-2, atol=0.5)

    def plan(h, **kw):
        return ops.contraction_plan(h, [M, K], "mk", [K, N], "kn", [M, N], "mn", dtype=cdt, workspace_limit=256 << 20, **kw)

    def run(p, ws):
        D.zero_()
        p.contract(1.0, A.data_ptr(), B.data_ptr(), 0.0, D.data_ptr(), D.data_ptr(), ws.data_ptr(), p.required_workspace)
        torch.cuda.synchronize()
        np.testing.assert_allclose(D.double().cpu().numpy(), ref, **tol)

    h = ops.Handle(plan_cache=128)                                           # contraction_plan_cache.cu:157-158
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    # what the first candidates are: plans by explicit rank, cache bypassed
    ranked = []
    for r in range(4):
        p = plan(h, algo=r, cache_mode=ct.CACHE_MODE_NONE)
        ranked.append(_choice(p.describe()))
        p.destroy()
    distinct = []
    for c in ranked:
        if c not in distinct:
            distinct.append(c)
    assert len(distinct) == expect_trials, ranked
    tune = dict(cache_mode=ct.CACHE_MODE_PEDANTIC, autotune=ct.AUTOTUNE_MODE_INCREMENTAL, incremental_count=4)
    seen = []
    for i in range(5):                                                       # :262 "last iteration will hit the cache"
        p = plan(h, **tune)
        seen.append(_choice(p.describe()))
        run(p, ws)
        p.destroy()
    assert seen[:expect_trials] == distinct, (seen, ranked)                  # trials walk the ranked candidates in order
    assert all(c in distinct for c in seen[expect_trials:]) and len(set(seen[expect_trials:])) == 1, seen
    tuned = seen[-1]
    # a plan without the autotune mode is answered from the cache as well
    p = plan(h)
    assert _choice(p.describe()) == tuned
    p.destroy()

    f = str(tmp_path / ("plancache_%s.txt" % dtype)).encode()
    assert ct.cutensorHandleWritePlanCacheToFile(h.h, f) == ct.STATUS_SUCCESS
    lines = open(f.decode()).read().splitlines()
    assert lines[0].startswith("cutensor-amd-plancache") and len(lines) == 2
    cols = lines[1].split("\t")
    assert (int(cols[1]), int(cols[2])) == tuned and int(cols[3]) == expect_trials and float(cols[4]) > 0.0

    # next process run: a fresh handle reads the file BEFORE it resizes its cache (:132-158)
    h2 = ops.Handle()
    n = ctypes.c_uint32(0)
    assert ct.cutensorHandleReadPlanCacheFromFile(h2.h, f, ctypes.byref(n)) == ct.STATUS_SUCCESS and n.value == 1
    ct.check(ct.cutensorHandleResizePlanCache(h2.h, 128))
    p = plan(h2)
    assert _choice(p.describe()) == tuned
    run(p, ws)
    p.destroy()
    p = plan(h2, **tune)                                                     # already tuned: no new trials
    assert _choice(p.describe()) == tuned
    p.destroy()
    # a cache that is too small for the file reports it (and keeps what fitted)
    h3 = ops.Handle()
    ct.check(ct.cutensorHandleResizePlanCache(h3.h, 0))
    st = ct.cutensorHandleReadPlanCacheFromFile(h3.h, f, ctypes.byref(n))
    assert st == ct.STATUS_INSUFFICIENT_WORKSPACE and n.value == 0
