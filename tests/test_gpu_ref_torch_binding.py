"""The reference's OWN numerical test on this path, run against OUR library on the GPU.

oracle/build_ref_torch_binding.sh compiles the reference's pybind11 module (cuTENSOR/python/cutensor/torch/einsum.cc over
cuTENSOR/python/einsum.h) UNMODIFIED against include/cutensor.h + lib/libcutensor.so and stages the reference's python
package (cutensor/common.py, cutensor/torch/einsum.py, cutensor/torch/einsum_test.py) around it under oracle/_ref/pyref/.
The staged tree travels to the GPU box with the snapshot; /root/reference is not read at run time.

What is pinned here is the reference's own assertion (einsum_test.py:35-42: rtol 5e-3 / atol 6e-3 against torch.einsum,
forward AND both gradients, seed 0) for its 10 binary cases (:47-124) and its 5 EinsumGeneral cases (:155-187), through
the reference's C++: Einsum<>::plan (einsum.h:277-399), ::execute (:411-442), the CUTENSOR_R_* spellings (:39-67), the
handle singleton (:455-502), cutensorPlanGetAttribute(REQUIRED_WORKSPACE) and the torch caching allocator for the
workspace (einsum.cc:96-123).  Nothing under cudalibrarysamples_amd/ imports the staged package: it is the checker."""
import os
import subprocess
import sys
import unittest

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PYREF = os.path.join(ROOT, "oracle", "_ref", "pyref")
PYSHIM = os.path.join(ROOT, "tests", "sample_compat", "pyshim")   # `parameterized` (not installed here): param + expand

# the names parameterized.expand generates for einsum_test.py's two lists, in file order
BINARY = ["0_test_0", "1_test_0_complex_", "2_test_1", "3_test_2", "4_test_3", "5_test_4", "6_test_5", "7_test_6", "8_test_7",
          "9_test_8"]
GENERAL = ["0_test_0", "1_test_1", "2_test_2", "3_test_3", "4_test_4"]


def _have():
    import glob
    return bool(glob.glob(os.path.join(PYREF, "cutensor", "torch", "binding*.so")))


@pytest.fixture(scope="module")
def ref_test_module(built):
    if not _have():
        pytest.skip("oracle/_ref/pyref was not built (reference tree absent at build time)")
    for p in (PYREF, PYSHIM):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cutensor.torch.einsum_test as mod   # the reference's file, staged verbatim
    import cutensor.torch.binding as binding   # the reference's einsum.cc, compiled verbatim
    assert os.path.realpath(binding.__file__).startswith(os.path.realpath(PYREF))
    return mod


def _production_counts():
    """Launch counters of the library instance(s) the reference's binding can be bound to: it is linked against "libcutensor.so" (rpath to
    lib/), which the dynamic loader satisfies with the flavour this process loaded first — the suite's lib_hooks/ (tests/conftest.py) —
    or, in a fresh process, with lib/.  Both instances' counters are added up."""
    import ctypes
    from cudalibrarysamples_amd import cutensor as ours
    total = ours.launch_counts()       # the flavour this process loaded first: a later DT_NEEDED "libcutensor.so" (same soname) binds to it
    lib = ctypes.CDLL(os.path.join(ROOT, "cudalibrarysamples_amd", "lib", "libcutensor.so"))
    if os.path.realpath(ours.LIB_PATH) != os.path.realpath(os.path.join(ROOT, "cudalibrarysamples_amd", "lib", "libcutensor.so")):
        out = (ctypes.c_uint64 * 5)()
        lib.ctamdLaunchCounts(out)
        for k, v in zip(("simple", "wide", "f32", "h16", "gen"), out):
            total[k] += int(v)
    return total


def _run(mod, name, expect_gen=False):
    """Runs one of the reference's test methods.  The reference's binding drives libcutensor.so directly, so which kernels its
    contractions ran on is read from the library's launch counters: never the scalar FMA fallback (gett_simple_kernel), and for
    the 16-bit / fp64 / complex cases the general MFMA family."""
    before = _production_counts()
    suite = unittest.defaultTestLoader.loadTestsFromName(name, mod.EinsumTest)
    assert suite.countTestCases() == 1, name
    res = unittest.TestResult()
    suite.run(res)
    problems = res.errors + res.failures
    assert not problems, problems[0][1]
    assert res.testsRun == 1 and not res.skipped
    after = _production_counts()
    delta = {k: after[k] - before[k] for k in after}
    assert delta["simple"] == 0, (name, delta)
    if expect_gen:          # (this also shows that the counters see the reference binding's launches: one library instance)
        # fp64 / complex: the general MFMA family; fp16: the LDS-DMA kernels when one contracted digit remains (round 6), else general
        assert delta["gen"] + delta["h16"] > 0, (name, delta)

# reference cases whose data type is not fp32 (einsum_test.py:55-68 complex, :84-115 fp16 / fp64): general MFMA family
NON_F32 = {"1_test_0_complex_", "2_test_1", "5_test_4", "6_test_5", "7_test_6", "8_test_7", "9_test_8"}


@pytest.mark.parametrize("case", BINARY)
def test_reference_einsum_equivalent_results(ref_test_module, case):
    """einsum_test.py:127-151 — EinsumFunction.apply forward + backward vs torch.einsum."""
    _run(ref_test_module, "test_einsum_equivalent_results_" + case, expect_gen=case in NON_F32)


@pytest.mark.parametrize("case", GENERAL)
def test_reference_einsum_general_equivalent_results(ref_test_module, case):
    """einsum_test.py:200-230 — EinsumGeneral (numpy einsum_path, pairwise contractions, unary reduction) fwd + bwd."""
    _run(ref_test_module, "test_einsum_general_equivalent_results_" + case)


def test_reference_case_list_is_complete(ref_test_module):
    """Every test the reference's file defines is named above — a case added upstream cannot be silently skipped."""
    names = sorted(n for n in dir(ref_test_module.EinsumTest)
                   if n.startswith("test_") and callable(getattr(ref_test_module.EinsumTest, n)))
    want = sorted(["test_einsum_equivalent_results_" + c for c in BINARY]
                  + ["test_einsum_general_equivalent_results_" + c for c in GENERAL])
    assert names == want


def test_reference_test_file_as_the_reference_runs_it(built):
    """`python einsum_test.py`-style: the file's own unittest.main() in a fresh interpreter (einsum_test.py:234-235)."""
    if not _have():
        pytest.skip("oracle/_ref/pyref was not built (reference tree absent at build time)")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PYREF, PYSHIM, os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-m", "unittest", "-v", "cutensor.torch.einsum_test"], capture_output=True, text=True,
                       timeout=900, env=env, cwd=PYREF)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "Ran 15 tests" in tail and "OK" in tail, tail


def test_reference_plan_execute_split(ref_test_module):
    """The binding's plan()/execute() pair (einsum.cc:148-233, einsum.py:69-90, module API :105-109): plan once, execute
    twice on refreshed inputs; not covered by the reference's test file, checked at its tolerance."""
    import torch
    import cutensor.torch as ct
    torch.manual_seed(0)
    a = torch.randn(20, 50, 50, 50, device="cuda")
    b = torch.randn(50, 50, 50, 20, device="cuda")
    m = ct.Einsum("mlik,lkjm->lij")
    p = m.plan(a, b)
    assert p.worksize >= 0 and p.workspace.numel() == p.worksize
    for _ in range(2):
        out = m.execute(p)
        torch.testing.assert_close(out, torch.einsum("mlik,lkjm->lij", a, b), rtol=5e-3, atol=6e-3)
        a.normal_()
        b.normal_()


def test_reference_binding_bf16(ref_test_module):
    """einsum_test.py:116-123 keeps the bf16 case commented out ("Activate when cuTENSOR supports it"); the binding carries
    the trait (einsum.cc:35-40, CUTENSOR_R_16BF + COMPUTE_DESC_16BF), so it is run here through the binding directly."""
    import torch
    import cutensor.torch as ct
    torch.manual_seed(0)
    a = torch.randn(20, 50, 50, 50, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(50, 50, 50, 20, device="cuda", dtype=torch.bfloat16)
    before = _production_counts()
    got = ct.EinsumFunction.apply("mlik,lkjm->lij", a, b)
    after = _production_counts()
    assert after["gen"] + after["h16"] > before["gen"] + before["h16"] and after["simple"] == before["simple"], (before, after)
    ref = torch.einsum("mlik,lkjm->lij", a.double(), b.double())
    # K = 20*50 terms of N(0,1) products: |ref| ~ 32; bf16 output rounding 2^-8 relative
    torch.testing.assert_close(got.double(), ref, rtol=1e-2, atol=0.25)


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_reference_binding_complex_unary_equations(ref_test_module, dtype):
    """Unary equations on complex tensors THROUGH THE REFERENCE'S UNMODIFIED BINDING: torch/einsum.cc:83 dispatches the complex types
    for every form, python/einsum.h:326-343 turns a one-operand equation into cutensorCreateReduction(OP_ADD) and :430-441 into
    cutensorReduce (einsum.cu:346-372) — a permutation ("ij->ji"), a reduction ("ijk->ik"), a full reduction, and EinsumGeneral with
    its unary step (einsum.py:139-150).  Against torch.einsum at the reference's own tolerance (einsum_test.py:35-42); the
    permutation also backward (its gradient is the inverse permutation; a reduction's gradient is a broadcast, which the
    reference's unary path does not express either)."""
    import torch
    import cutensor.torch as ct
    tdt = getattr(torch, dtype)
    torch.manual_seed(0)
    z2 = torch.randn(50, 64, dtype=tdt, device="cuda", requires_grad=True)
    z3 = torch.randn(20, 50, 30, dtype=tdt, device="cuda")
    tol = dict(rtol=5e-3, atol=6e-3)
    out = ct.EinsumFunction.apply("ij->ji", z2)
    torch.testing.assert_close(out, torch.einsum("ij->ji", z2.detach()), **tol)
    g = torch.randn_like(out)
    out.backward(g)
    torch.testing.assert_close(z2.grad, g.t().contiguous(), **tol)
    for eq in ("ijk->ik", "ijk->kji", "ijk->", "ijk->j"):
        torch.testing.assert_close(ct.EinsumFunction.apply(eq, z3), torch.einsum(eq, z3), **tol)
    torch.testing.assert_close(ct.EinsumGeneral("ijk->ki", z3), torch.einsum("ijk->ki", z3), **tol)
    # a binary step followed by nothing unary, for completeness of the N-ary path on complex data
    w = torch.randn(30, 8, dtype=tdt, device="cuda")
    torch.testing.assert_close(ct.EinsumGeneral("ijk,kl->ijl", z3, w), torch.einsum("ijk,kl->ijl", z3, w), **tol)
