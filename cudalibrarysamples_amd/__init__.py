"""cudalibrarysamples_amd — MI355X-native tensor-contraction engine behind the cuTENSOR C ABI.

The product is `lib/libcutensor.so` (C ABI declared in include/cutensor.h, gfx950 HIP kernels in
csrc/kernels).  This package is the thin Python host side: it loads the library, mirrors the ABI
one-to-one (`cudalibrarysamples_amd.cutensor`) and offers the reference binding's einsum entry point
(`cudalibrarysamples_amd.torch_einsum`, after cuTENSOR/python/cutensor/torch/einsum.py).

There is no CPU fallback: importing `cutensor` raises if the HIP library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C cudalibrarysamples_amd/csrc`).
"""
from . import cutensor  # noqa: F401

__all__ = ["cutensor"]
