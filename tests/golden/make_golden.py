"""Generates tests/golden/*.npz: inputs and torch.einsum (CPU) outputs for the test cases of the
reference's own numerical test file, cuTENSOR/python/cutensor/torch/einsum_test.py:47-124
(same equations, dtypes, seed = torch.manual_seed(0), randn inputs).  The reference binding itself
needs CUDA + the closed libcutensor and cannot be imported here, so the fixtures hold what the
reference's test compares against: torch.einsum.  Extents 50 are kept where the file stays small and
shrunk to 20 (marked in meta) otherwise, so that the committed fixtures total < 2 MB.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # (name, a_size, b_size, equation, dtype, shrunk)
    ("test0_f32", (48, 37), (37, 74), "ik,kj->ij", torch.float32, False),
    ("test2_f32", (20, 20, 20, 8), (20, 20, 20, 8), "likm,lkjm->lij", torch.float32, True),
    ("test3_f32", (8, 20, 20, 20), (20, 20, 20, 8), "mlik,lkjm->lij", torch.float32, True),
    ("test4_f16", (50, 50), (50, 50), "ik,kj->ij", torch.float16, False),
    ("test5_f16", (20, 20, 20), (20, 20, 20), "lik,lkj->lij", torch.float16, True),
    ("test7_f16", (8, 20, 20, 20), (20, 20, 20, 8), "mlik,lkjm->lij", torch.float16, True),
    ("test8_f64", (2, 5, 50, 2), (5, 2, 50, 2), "mlik,lkjm", torch.float64, False),
    ("test8_bf16", (8, 20, 20, 20), (20, 20, 20, 8), "mlik,lkjm->lij", torch.bfloat16, True),
]


def main():
    for name, a_size, b_size, eq, dtype, shrunk in CASES:
        torch.manual_seed(0)
        a = torch.randn(*a_size, dtype=torch.float32).to(dtype)
        b = torch.randn(*b_size, dtype=torch.float32).to(dtype)
        # reference value in fp64 from the (rounded) inputs: what torch.einsum converges to
        out = torch.einsum(eq, a.double(), b.double())
        store = np.float64 if dtype == torch.float64 else np.float32
        meta = dict(equation=eq, dtype=str(dtype).replace("torch.", ""), shrunk=shrunk,
                    source="cuTENSOR/python/cutensor/torch/einsum_test.py:47-124")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), a=a.double().numpy().astype(store),
                            b=b.double().numpy().astype(store), out=out.numpy().astype(store),
                            meta=json.dumps(meta))
        print(name, tuple(out.shape))


if __name__ == "__main__":
    main()
