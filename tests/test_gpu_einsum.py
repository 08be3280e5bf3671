"""GPU parity of the einsum front end against the golden fixtures of the reference's own test cases
(tests/golden: the parameter list parsed out of cuTENSOR/python/cutensor/torch/einsum_test.py:45-125 by
tests/golden/make_golden.py, every dtype of it — fp32, fp64, fp16, complex64, complex128 and the commented-out bf16
case) at the reference's tolerance (rtol 5e-3, atol 6e-3; :35-42) — and tighter for fp32/fp64/complex."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def te(built):
    import torch
    assert torch.cuda.is_available()
    from cudalibrarysamples_amd import torch_einsum
    return torch, torch_einsum


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz")))
def test_golden_cases(te, name):
    torch, torch_einsum = te
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    dtype = getattr(torch, meta["dtype"])
    a = torch.from_numpy(z["a"]).to(dtype).cuda()
    b = torch.from_numpy(z["b"]).to(dtype).cuda()
    out = torch_einsum.einsum(meta["equation"], a, b)
    torch.cuda.synchronize()
    assert out.dtype == dtype
    _assert_mfma_kernel(torch_einsum, meta["equation"], a, b)
    if dtype.is_complex:
        got, ref = out.to(torch.complex128).cpu().numpy(), z["out"].astype(np.complex128)
        assert list(got.shape) == list(ref.shape)
        for part in (np.real, np.imag):     # einsum_test.py:38-40
            np.testing.assert_allclose(part(got), part(ref), rtol=5e-3, atol=6e-3)
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
        return
    got = out.double().cpu().numpy()
    ref = z["out"].astype(np.float64)
    assert list(got.shape) == list(ref.shape)
    np.testing.assert_allclose(got, ref, rtol=5e-3, atol=6e-3)
    if meta["dtype"] in ("float32", "float64"):
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)


def _assert_mfma_kernel(torch_einsum, equation, a, b):
    """Green parity on a fallback is weaker evidence than it looks: every golden case must have run on a matrix-core kernel —
    fp32 on the fp32 MFMA families, fp64 and complex on the general MFMA family, 16-bit data on the LDS-DMA kernels when ONE contracted
    digit remains (round 6: these extents of 50 have no 16-byte lanes and a partial k-unit — the masked / repaired last K-tile) and on
    the general family otherwise, none on gett_simple_kernel / gett_wide_kernel."""
    d = torch_einsum._plans[(equation, tuple(a.shape), tuple(b.shape), a.dtype, False, False)].describe()
    assert d["kernel"] >= 0 and d["kname"] not in ("gett_simple_kernel", "gett_wide_kernel"), d
    if str(a.dtype) == "torch.float32":
        assert d["family"] == 0, d
    elif str(a.dtype) in ("torch.float16", "torch.bfloat16"):
        assert d["family"] == (1 if len(d["Kdigits"]) == 1 else 2), d
    else:
        assert d["family"] == 2, d


def test_demo_equations(te):
    """einsum.cu:447-451 (shapes {2,4,5},{4,8,7}) with real data, against numpy.einsum."""
    torch, torch_einsum = te
    rng = np.random.default_rng(5)
    a = rng.random((2, 4, 5), dtype=np.float32)
    b = rng.random((4, 8, 7), dtype=np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    for eq in ("ijn,jmk->inkm", "ijn,jmk"):
        got = torch_einsum.einsum(eq, ta, tb).cpu().numpy()
        np.testing.assert_allclose(got, np.einsum(eq, a, b), rtol=1e-5)
    for eq in ("nij", "nij->ijn", "nij->ji"):
        got = torch_einsum.einsum(eq, ta).cpu().numpy()
        np.testing.assert_allclose(got, np.einsum(eq, a), rtol=1e-5)


def test_unsupported_equation_raises(te):
    torch, torch_einsum = te
    a = torch.zeros(2, 3, 4, device="cuda")
    with pytest.raises(ValueError):
        torch_einsum.einsum("ab...->a", a)


FULL = os.path.join(GOLDEN, "full")


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(FULL) if f.endswith(".npz")))
def test_golden_cases_at_the_reference_extents(te, name):
    """The reference's cases at ITS sizes (extents of 50, einsum_test.py:47-124): inputs redrawn as tests/golden/make_golden.py drew
    them (probe-checked), the engine's output compared at the fixture's 4096 sampled positions and through sum |out|."""
    from tests.util import golden_inputs
    torch, torch_einsum = te
    z = np.load(os.path.join(FULL, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    a, b = golden_inputs(meta["dtype"], meta["a_size"], meta["b_size"])
    wide = torch.complex128 if a.dtype.is_complex else torch.float64
    np.testing.assert_array_equal(a.to(wide).numpy().reshape(-1)[:16], z["a_probe"])
    np.testing.assert_array_equal(b.to(wide).numpy().reshape(-1)[:16], z["b_probe"])
    out = torch_einsum.einsum(meta["equation"], a.cuda(), b.cuda())
    torch.cuda.synchronize()
    assert out.dtype == a.dtype and list(out.shape) == meta["out_shape"]
    _assert_mfma_kernel(torch_einsum, meta["equation"], a, b)
    full = out.to(wide).cpu().numpy()
    got, ref = full.reshape(-1)[z["idx"]], z["out_sampled"]
    for part in ((np.real, np.imag) if np.iscomplexobj(ref) else (lambda x: x,)):
        np.testing.assert_allclose(part(got), part(ref), rtol=5e-3, atol=6e-3)      # einsum_test.py:35-42
    if meta["dtype"] in ("float32", "float64", "complex64", "complex128"):
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(np.abs(full).sum(), float(z["sum_abs"]), rtol=2e-3 if meta["dtype"] in ("float16", "bfloat16") else 1e-5)
