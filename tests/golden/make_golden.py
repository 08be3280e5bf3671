"""Generates tests/golden/*.npz: inputs and torch.einsum (CPU) outputs for the test cases of the reference's own
numerical test file, cuTENSOR/python/cutensor/torch/einsum_test.py.

The case list is READ FROM THAT FILE: the `param(...)` calls inside the `@parameterized.expand([...])` decorator of
`EinsumTest.test_einsum_equivalent_results` (:45-125) are parsed with `ast` — name, a_size, b_size, equation, dtype —
without importing the module (it needs CUDA and the closed libcutensor).  The bf16 case the reference keeps commented
out ("Activate when cuTENSOR supports it", :115-123) is recovered from the comment block the same way.  Inputs follow
the test body (:131-147): torch.manual_seed(0), randn.  The reference binding itself cannot run here, so the fixtures
hold what its test compares against: torch.einsum, evaluated in fp64 (complex128) on the inputs rounded to the case's
dtype.  Extents of 50 are shrunk to 20 (or 10) — recorded in meta together with the source line of the case — wherever an
operand would exceed 40,000 elements, so that the committed fixtures stay small (< 2 MB in total).

tests/golden/full/*.npz pin the same cases AT THE REFERENCE'S OWN EXTENTS (50): the inputs are not stored (they are redrawn by
tests/util.py golden_inputs(), the same lines as here; 16 leading values of each operand are stored as a drift probe), the
output is stored at 4096 fixed random positions together with sum |out| over the whole tensor.

Run from the repo root, in the build container (needs /root/reference):  python tests/golden/make_golden.py
"""
import ast
import json
import os
import re

import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import golden_inputs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF_TEST = "/root/reference/cuTENSOR/python/cutensor/torch/einsum_test.py"
MAX_ELEMS = 40_000       # per operand: extents of 50 shrink to 20, and to 10 if that is still too much


def _param_calls(tree):
    """param(...) nodes inside the parameterized.expand list of test_einsum_equivalent_results."""
    for cls in ast.walk(tree):
        if not isinstance(cls, ast.ClassDef) or cls.name != "EinsumTest":
            continue
        for fn in cls.body:
            if isinstance(fn, ast.FunctionDef) and fn.name == "test_einsum_equivalent_results":
                for dec in fn.decorator_list:
                    for node in ast.walk(dec):
                        if isinstance(node, ast.Call) and getattr(node.func, "id", "") == "param":
                            yield node


def _case(node, lineno=None):
    kw = {k.arg: k.value for k in node.keywords}
    dtype = ast.unparse(kw["dtype"]).replace("torch.", "")
    return dict(name=ast.literal_eval(node.args[0]), a_size=ast.literal_eval(kw["a_size"]), b_size=ast.literal_eval(kw["b_size"]),
                equation=ast.literal_eval(kw["equation"]), dtype=dtype, line=lineno or node.lineno)


def reference_cases():
    src = open(REF_TEST).read()
    cases = [_case(n) for n in _param_calls(ast.parse(src))]
    # the commented-out case: a run of '# ' lines that holds a param( ... ) call
    lines = src.splitlines()
    block, start = [], None
    for i, ln in enumerate(lines, 1):
        m = re.match(r"\s*#\s?(.*)$", ln)
        if m and ("param(" in m.group(1) or block):
            if not block:
                start = i
            block.append(m.group(1))
            if m.group(1).strip().startswith("),") or m.group(1).strip() == ")":
                try:
                    node = ast.parse("\n".join(block).strip().rstrip(",")).body[0].value
                    c = _case(node, start)
                    c["commented_out_in_reference"] = True
                    cases.append(c)
                except SyntaxError:
                    pass
                block = []
        elif block:
            block = []
    return cases


def shrink_to(sizes):
    """Replacement for extent 50 that brings every operand of the case under MAX_ELEMS (None = keep 50)."""
    for repl in (None, 20, 10):
        cand = [tuple((repl if (repl and e == 50) else e) for e in s) for s in sizes]
        if all(int(np.prod(s)) <= MAX_ELEMS for s in cand):
            return cand, repl
    return cand, repl


def main():
    cases = reference_cases()
    assert len(cases) >= 11, cases      # 10 live cases + the commented-out bf16 one
    for old in os.listdir(HERE):
        if old.endswith(".npz"):
            os.remove(os.path.join(HERE, old))
    for c in cases:
        tdt = getattr(torch, c["dtype"])
        (a_size, b_size), repl = shrink_to([c["a_size"], c["b_size"]])   # both operands alike: they share modes
        sa = sb = repl is not None
        torch.manual_seed(0)
        if tdt.is_complex:
            a = torch.randn(*a_size, dtype=tdt)
            b = torch.randn(*b_size, dtype=tdt)
            wide, store = torch.complex128, (np.complex64 if tdt == torch.complex64 else np.complex128)
        else:
            a = torch.randn(*a_size, dtype=torch.float32).to(tdt)
            b = torch.randn(*b_size, dtype=torch.float32).to(tdt)
            wide, store = torch.float64, (np.float64 if tdt == torch.float64 else np.float32)
        out = torch.einsum(c["equation"], a.to(wide), b.to(wide))      # what torch.einsum converges to
        name = re.sub(r"[^a-z0-9]+", "_", c["name"].lower()).strip("_") + "_" + c["dtype"]
        meta = dict(equation=c["equation"], dtype=c["dtype"], shrunk=bool(sa or sb), extent_50_became=repl, a_size=list(c["a_size"]), b_size=list(c["b_size"]),
                    reference_case=c["name"], commented_out_in_reference=bool(c.get("commented_out_in_reference")),
                    source="cuTENSOR/python/cutensor/torch/einsum_test.py:%d" % c["line"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), a=a.to(wide).numpy().astype(store), b=b.to(wide).numpy().astype(store),
                            out=out.numpy().astype(store), meta=json.dumps(meta))
        print(name, a_size, b_size, tuple(out.shape), meta["source"])
        # ---- the same case at the reference's own extents: sampled outputs + a norm ----
        fa, fb = golden_inputs(c["dtype"], c["a_size"], c["b_size"])
        fout = torch.einsum(c["equation"], fa.to(wide), fb.to(wide)).numpy()
        flat = fout.reshape(-1)
        rng = np.random.default_rng(20250922)
        idx = np.sort(rng.choice(flat.size, size=min(4096, flat.size), replace=False)).astype(np.int64)
        fmeta = dict(meta, shrunk=False, extent_50_became=None, out_shape=list(fout.shape))
        os.makedirs(os.path.join(HERE, "full"), exist_ok=True)
        np.savez_compressed(os.path.join(HERE, "full", name + ".npz"), idx=idx, out_sampled=flat[idx], sum_abs=np.float64(np.abs(flat).sum()),
                            a_probe=fa.to(wide).numpy().reshape(-1)[:16], b_probe=fb.to(wide).numpy().reshape(-1)[:16], meta=json.dumps(fmeta))


if __name__ == "__main__":
    main()
