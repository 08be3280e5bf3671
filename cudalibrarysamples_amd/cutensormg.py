"""ctypes mirror of include/cutensorMg.h (call sites: cuTENSORMg/contraction_multi_gpu.cu:151-383)."""
import ctypes
import os

from . import cutensor as ct   # loads libcutensor.so with RTLD_GLOBAL first

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libcutensorMg.so")
if not os.path.exists(LIB_PATH):
    raise ImportError("libcutensorMg.so is not built: %s missing (no CPU fallback)" % LIB_PATH)
lib = ctypes.CDLL(LIB_PATH)

COMPUTE_32F = 1 << 2
ALGO_DEFAULT = -1
_vp = ctypes.c_void_p
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_vpp = ctypes.POINTER(ctypes.c_void_p)

EXPORTS = {
    "cutensorMgCreate": (ctypes.POINTER(_vp), ctypes.c_uint32, _i32p),
    "cutensorMgDestroy": (_vp,),
    "cutensorMgCreateTensorDescriptor": (_vp, ctypes.POINTER(_vp), ctypes.c_uint32, _i64p, _i64p, _i64p, _i64p, _i32p,
                                         ctypes.c_uint32, _i32p, ctypes.c_int),
    "cutensorMgDestroyTensorDescriptor": (_vp,),
    "cutensorMgCreateContractionDescriptor": (_vp, ctypes.POINTER(_vp), _vp, _i32p, _vp, _i32p, _vp, _i32p, _vp, _i32p, ctypes.c_int),
    "cutensorMgDestroyContractionDescriptor": (_vp,),
    "cutensorMgCreateContractionFind": (_vp, ctypes.POINTER(_vp), ctypes.c_int),
    "cutensorMgDestroyContractionFind": (_vp,),
    "cutensorMgContractionGetWorkspace": (_vp, _vp, _vp, ctypes.c_int, _i64p, _i64p),
    "cutensorMgCreateContractionPlan": (_vp, ctypes.POINTER(_vp), _vp, _vp, _i64p, ctypes.c_int64),
    "cutensorMgDestroyContractionPlan": (_vp,),
    "cutensorMgContraction": (_vp, _vp, _vp, _vpp, _vpp, _vp, _vpp, _vpp, _vpp, _vp, _vpp),
}
for _name, _args in EXPORTS.items():
    _f = getattr(lib, _name)
    _f.argtypes = list(_args)
    _f.restype = ctypes.c_int
    globals()[_name] = _f

check = ct.check
i64, i32 = ct.i64, ct.i32


def ptr_array(values):
    return (ctypes.c_void_p * max(len(values), 1))(*values)
