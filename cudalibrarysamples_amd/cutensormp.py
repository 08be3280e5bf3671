"""ctypes mirror of include/cutensorMp.h (call sites: cutensorMp/cutensorMp_contraction.cu:470-590).

`LocalWorld` / `run_ranks` drive several ranks as threads of this process on one GPU (the library's in-process
exchange layer) — the way the multi-rank path is exercised on a single-GPU machine.  With real multi-GPU jobs the
handle is created from an RCCL communicator instead (`cutensorMpCreate`)."""
import ctypes
import json
import os
import threading

from . import cutensor as ct   # loads libcutensor.so with RTLD_GLOBAL first

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib_hooks" if os.environ.get("CTAMD_LIB_FLAVOUR") == "hooks" else "lib", "libcutensorMp.so")
if not os.path.exists(LIB_PATH):
    raise ImportError("libcutensorMp.so is not built: %s missing (no CPU fallback)" % LIB_PATH)
lib = ctypes.CDLL(LIB_PATH)

ALGO_DEFAULT = -1
PLAN_REQUIRED_WORKSPACE_DEVICE = 0
PLAN_REQUIRED_WORKSPACE_HOST = 1
_vp = ctypes.c_void_p
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)

EXPORTS = {
    "cutensorMpCreate": (ctypes.POINTER(_vp), _vp, ctypes.c_int, _vp),
    "cutensorMpDestroy": (_vp,),
    "cutensorMpCreateTensorDescriptor": (_vp, ctypes.POINTER(_vp), ctypes.c_uint32, _i64p, _i64p, _i64p, _i64p, _i64p,
                                         ctypes.c_uint32, _i32p, ctypes.c_int),
    "cutensorMpDestroyTensorDescriptor": (_vp,),
    "cutensorMpCreateContraction": (_vp, ctypes.POINTER(_vp), _vp, _i32p, ctypes.c_int, _vp, _i32p, ctypes.c_int,
                                    _vp, _i32p, ctypes.c_int, _vp, _i32p, _vp),
    "cutensorMpDestroyOperationDescriptor": (_vp,),
    "cutensorMpCreatePlanPreference": (_vp, ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64),
    "cutensorMpDestroyPlanPreference": (_vp,),
    "cutensorMpCreatePlan": (_vp, ctypes.POINTER(_vp), _vp, _vp),
    "cutensorMpPlanGetAttribute": (_vp, _vp, ctypes.c_int, _vp, ctypes.c_size_t),
    "cutensorMpDestroyPlan": (_vp,),
    "cutensorMpContract": (_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp),
    "ctamdMpLocalWorldCreate": (ctypes.POINTER(_vp), ctypes.c_int),
    "ctamdMpLocalWorldDestroy": (_vp,),
    "ctamdMpCreateOnLocalWorld": (ctypes.POINTER(_vp), _vp, ctypes.c_int, ctypes.c_int, _vp),
}
for _name, _args in EXPORTS.items():
    _f = getattr(lib, _name)
    _f.argtypes = list(_args)
    _f.restype = ctypes.c_int
    globals()[_name] = _f
lib.ctamdMpDescribePlan.argtypes = [_vp, ctypes.c_char_p, ctypes.c_size_t]
lib.ctamdMpDescribePlan.restype = ctypes.c_size_t

check = ct.check
i64, i32 = ct.i64, ct.i32


def describe_plan(plan):
    n = lib.ctamdMpDescribePlan(plan, None, 0)
    buf = ctypes.create_string_buffer(n)
    lib.ctamdMpDescribePlan(plan, buf, n)
    return json.loads(buf.value.decode())


class LocalWorld:
    def __init__(self, nranks):
        self.nranks = nranks
        self.ptr = _vp()
        check(lib.ctamdMpLocalWorldCreate(ctypes.byref(self.ptr), nranks))

    def close(self):
        if self.ptr:
            lib.ctamdMpLocalWorldDestroy(self.ptr)
            self.ptr = _vp()


def run_ranks(nranks, fn):
    """Runs fn(rank) on `nranks` threads at once and returns the list of results; re-raises the first exception."""
    out, err = [None] * nranks, [None] * nranks

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:   # noqa: BLE001 — reported to the caller below
            err[r] = e

    threads = [threading.Thread(target=body, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out
