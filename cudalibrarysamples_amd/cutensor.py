"""ctypes mirror of include/cutensor.h — same names, same argument meaning, same status codes.

Reference interface being mirrored: the cuTENSOR 2.x call sequence of cuTENSOR/contraction.cu:122-265,
cuTENSOR/reduction.cu:141-222 and cuTENSOR/elementwise_permute.cu:142-200.  Functions return the raw
cutensorStatus_t; `check()` turns a non-success status into CuTensorError (the samples' handle_error,
cuTENSOR/utils.cuh:35-39).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUTENSOR_AMD_LIBRARY: another build of the same library (A/B measurements of two source revisions on one box)
# CTAMD_LIB_FLAVOUR=hooks: the test-hooks flavour (lib_hooks/, make HOOKS=1) — the same kernels, host code that reads the test / measurement
# switches (csrc/host/api_guard.hpp); tests/conftest.py and the tools that drive such switches select it, bench.py and smoke() never do
LIB_DIR = "lib_hooks" if os.environ.get("CTAMD_LIB_FLAVOUR") == "hooks" else "lib"
LIB_PATH = os.environ.get("CUTENSOR_AMD_LIBRARY") or os.path.join(_HERE, LIB_DIR, "libcutensor.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libcutensor.so (gfx950 HIP engine) is not built: %s missing. Build it with "
        "`make -C cudalibrarysamples_amd/csrc` or `__graft_entry__.build()`; there is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
if LIB_DIR == "lib" and not os.environ.get("CUTENSOR_AMD_LIBRARY"):
    _ignored = sorted(k for k in os.environ if k.startswith(("CUTENSOR_AMD_", "CUTENSORMP_AMD_")) and k != "CUTENSOR_AMD_LIBRARY")
    if _ignored:
        import warnings
        warnings.warn("the production libcutensor.so does not read %s: set CTAMD_LIB_FLAVOUR=hooks to load the test-hooks flavour" % ", ".join(_ignored))

# ---- enums (include/cutensor/types.h) ----------------------------------------------------------
R_32F, R_64F, R_16F, R_16BF = 0, 1, 2, 14
C_32F, C_64F = 4, 5   # complex data: contractions only (general MFMA family; mode-table kernel beyond four modes per group)
STATUS_SUCCESS, STATUS_NOT_INITIALIZED, STATUS_INVALID_VALUE = 0, 1, 7
STATUS_NOT_SUPPORTED, STATUS_INSUFFICIENT_WORKSPACE, STATUS_IO_ERROR = 15, 19, 21
OP_IDENTITY, OP_ADD, OP_MUL, OP_MAX, OP_MIN = 1, 3, 5, 6, 7
OP_CONJ = 9
ALGO_DEFAULT, ALGO_DEFAULT_PATIENT = -1, -6
WORKSPACE_MIN, WORKSPACE_DEFAULT, WORKSPACE_MAX = 1, 2, 3
JIT_MODE_NONE = 0
OPERATION_DESCRIPTOR_TAG, OPERATION_DESCRIPTOR_SCALAR_TYPE, OPERATION_DESCRIPTOR_FLOPS, OPERATION_DESCRIPTOR_MOVED_BYTES = 0, 1, 2, 3
PLAN_REQUIRED_WORKSPACE = 0
PLAN_PREFERENCE_AUTOTUNE_MODE, PLAN_PREFERENCE_CACHE_MODE, PLAN_PREFERENCE_INCREMENTAL_COUNT = 0, 1, 2
PLAN_PREFERENCE_ALGO, PLAN_PREFERENCE_KERNEL_RANK = 3, 4
AMD_PLAN_PREFERENCE_OPERANDS_STREAMED = 1000     # engine extension (include/cutensor/types.h): operands come from HBM on every call
AUTOTUNE_MODE_NONE, AUTOTUNE_MODE_INCREMENTAL = 0, 1
CACHE_MODE_NONE, CACHE_MODE_PEDANTIC = 0, 1

_vp = ctypes.c_void_p
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)

EXPORTS = {
    # name: (argtypes)
    "cutensorCreate": (ctypes.POINTER(_vp),),
    "cutensorDestroy": (_vp,),
    "cutensorHandleResizePlanCache": (_vp, ctypes.c_uint32),
    "cutensorHandleWritePlanCacheToFile": (_vp, ctypes.c_char_p),
    "cutensorHandleReadPlanCacheFromFile": (_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32)),
    "cutensorCreateTensorDescriptor": (_vp, ctypes.POINTER(_vp), ctypes.c_uint32, _i64p, _i64p, ctypes.c_int, ctypes.c_uint32),
    "cutensorDestroyTensorDescriptor": (_vp,),
    "cutensorCreateContraction": (_vp, ctypes.POINTER(_vp), _vp, _i32p, ctypes.c_int, _vp, _i32p, ctypes.c_int,
                                  _vp, _i32p, ctypes.c_int, _vp, _i32p, _vp),
    "cutensorCreateReduction": (_vp, ctypes.POINTER(_vp), _vp, _i32p, ctypes.c_int, _vp, _i32p, ctypes.c_int,
                                _vp, _i32p, ctypes.c_int, _vp),
    "cutensorCreatePermutation": (_vp, ctypes.POINTER(_vp), _vp, _i32p, ctypes.c_int, _vp, _i32p, _vp),
    "cutensorCreateElementwiseBinary": (_vp, ctypes.POINTER(_vp), _vp, _i32p, ctypes.c_int, _vp, _i32p, ctypes.c_int,
                                        _vp, _i32p, ctypes.c_int, _vp),
    "cutensorDestroyOperationDescriptor": (_vp,),
    "cutensorOperationDescriptorGetAttribute": (_vp, _vp, ctypes.c_int, _vp, ctypes.c_size_t),
    "cutensorOperationDescriptorSetAttribute": (_vp, _vp, ctypes.c_int, _vp, ctypes.c_size_t),
    "cutensorCreatePlanPreference": (_vp, ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int),
    "cutensorDestroyPlanPreference": (_vp,),
    "cutensorPlanPreferenceSetAttribute": (_vp, _vp, ctypes.c_int, _vp, ctypes.c_size_t),
    "cutensorEstimateWorkspaceSize": (_vp, _vp, _vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)),
    "cutensorCreatePlan": (_vp, ctypes.POINTER(_vp), _vp, _vp, ctypes.c_uint64),
    "cutensorDestroyPlan": (_vp,),
    "cutensorPlanGetAttribute": (_vp, _vp, ctypes.c_int, _vp, ctypes.c_size_t),
    "cutensorContract": (_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp),
    "cutensorReduce": (_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp),
    "cutensorPermute": (_vp, _vp, _vp, _vp, _vp, _vp),
    "cutensorElementwiseBinaryExecute": (_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp),
    "cutensorCreateElementwiseTrinary": (_vp, ctypes.POINTER(_vp), _vp, _i32p, ctypes.c_int, _vp, _i32p, ctypes.c_int,
                                         _vp, _i32p, ctypes.c_int, _vp, _i32p, ctypes.c_int, ctypes.c_int, _vp),
    "cutensorElementwiseTrinaryExecute": (_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp),
    "cutensorCreateContractionTrinary": (_vp, ctypes.POINTER(_vp), _vp, _i32p, ctypes.c_int, _vp, _i32p, ctypes.c_int,
                                         _vp, _i32p, ctypes.c_int, _vp, _i32p, ctypes.c_int, _vp, _i32p, _vp),
    "cutensorContractTrinary": (_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _vp),
    "cutensorCreateBlockSparseTensorDescriptor": (_vp, ctypes.POINTER(_vp), ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32),
                                                  _i64p, _i32p, _i64p, ctypes.c_int),
    "cutensorDestroyBlockSparseTensorDescriptor": (_vp,),
    "cutensorCreateBlockSparseContraction": (_vp, ctypes.POINTER(_vp), _vp, _i32p, ctypes.c_int, _vp, _i32p, ctypes.c_int,
                                             _vp, _i32p, ctypes.c_int, _vp, _i32p, _vp),
    "cutensorBlockSparseContract": (_vp, _vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                    _vp, ctypes.c_uint64, _vp),
    "cutensorReadKernelCacheFromFile": (_vp, ctypes.c_char_p),
    "cutensorWriteKernelCacheToFile": (_vp, ctypes.c_char_p),
}
DATA_SYMBOLS = ["CUTENSOR_COMPUTE_DESC_16F", "CUTENSOR_COMPUTE_DESC_16BF", "CUTENSOR_COMPUTE_DESC_TF32",
                "CUTENSOR_COMPUTE_DESC_3XTF32", "CUTENSOR_COMPUTE_DESC_32F", "CUTENSOR_COMPUTE_DESC_64F"]

for _name, _args in EXPORTS.items():
    _f = getattr(lib, _name)
    _f.argtypes = list(_args)
    _f.restype = ctypes.c_int
    globals()[_name] = _f
lib.cutensorGetErrorString.argtypes = [ctypes.c_int]
lib.cutensorGetErrorString.restype = ctypes.c_char_p
lib.cutensorGetVersion.restype = ctypes.c_size_t
lib.ctamdDescribePlan.argtypes = [_vp, ctypes.c_char_p, ctypes.c_size_t]
lib.ctamdCountCandidates.argtypes = [_vp, _vp, ctypes.c_uint64]
lib.ctamdPlanMemoStats.argtypes = [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
lib.ctamdPlanMemoStats.restype = None
lib.ctamdLaunchCounts.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
lib.ctamdLaunchCounts.restype = None
lib.ctamdLastH16Kernel.argtypes = []
lib.ctamdLastH16Kernel.restype = ctypes.c_int
lib.ctamdSetTimingBuffer.argtypes = [_vp, _vp]
lib.ctamdSetTimingBuffer.restype = None
lib.ctamdSetSplitKFold.argtypes = [_vp, ctypes.c_int]
lib.ctamdSetSplitKFold.restype = None
lib.ctamdProfileBegin.argtypes = [_vp]
lib.ctamdProfileBegin.restype = None
lib.ctamdProfileEnd.argtypes = [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
lib.ctamdEinsumCreate.argtypes = [ctypes.c_char_p, _i64p, ctypes.c_int, _i64p, ctypes.c_int, ctypes.c_int]
lib.ctamdEinsumCreate.restype = _vp
lib.ctamdEinsumSetConjugate.argtypes = [_vp, ctypes.c_int, ctypes.c_int]
lib.ctamdEinsumSetConjugate.restype = None
lib.ctamdEinsumDestroy.argtypes = [_vp]
lib.ctamdEinsumDestroy.restype = None
lib.ctamdEinsumIsInitialized.argtypes = [_vp]
lib.ctamdEinsumOutputShape.argtypes = [_vp, _i64p, ctypes.c_int]
lib.ctamdEinsumPlan.argtypes = [_vp, _vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
lib.ctamdEinsumReplan.argtypes = [_vp, _vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
lib.ctamdEinsumReplan.restype = ctypes.c_int
lib.ctamdMeasureMfmaCeiling.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
lib.ctamdMeasureMfmaCeiling.restype = ctypes.c_int
lib.ctamdMeasureMfmaCeilingShape.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
lib.ctamdMeasureMfmaCeilingShape.restype = ctypes.c_int
lib.ctamdEinsumExecute.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp]
lib.ctamdEinsumRawPlan.argtypes = [_vp]
lib.ctamdEinsumRawPlan.restype = _vp


def plan_memo_stats(handle):
    """(hits, misses, entries) of the handle's plan memo (cutensorCreatePlan answered by cloning a prototype / by planning)."""
    h, m, e = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint32(0)
    lib.ctamdPlanMemoStats(handle, ctypes.byref(h), ctypes.byref(m), ctypes.byref(e))
    return h.value, m.value, e.value


def launch_counts():
    """cutensorContract launches by kernel kind since the library was loaded: dict with the keys simple (scalar FMA fallback),
    wide (mode-table kernel), f32 (fp32 MFMA families), h16 (aligned 16-bit MFMA family), gen (general MFMA family)."""
    out = (ctypes.c_uint64 * 5)()
    lib.ctamdLaunchCounts(out)
    return dict(zip(("simple", "wide", "f32", "h16", "gen"), [int(v) for v in out]))


def last_h16_kernel():
    """Table entry of the kernel the last cutensorContract of the aligned 16-bit family launched (-1: none yet): 48..55 the one-tile
    256 x 256 kernel, 88..95 its persistent form — which twin ran is decided at the call (beta, the layout of C)."""
    return int(lib.ctamdLastH16Kernel())


def compute_desc(name):
    """Value of the exported data symbol CUTENSOR_COMPUTE_DESC_<name> (an opaque pointer)."""
    return _vp.in_dll(lib, "CUTENSOR_COMPUTE_DESC_" + name)


class CuTensorError(RuntimeError):
    def __init__(self, status):
        self.status = status
        RuntimeError.__init__(self, lib.cutensorGetErrorString(status).decode())


def check(status):
    if status != STATUS_SUCCESS:
        raise CuTensorError(status)


def getErrorString(status):
    return lib.cutensorGetErrorString(status).decode()


def describe_plan(plan):
    buf = ctypes.create_string_buffer(1024)
    lib.ctamdDescribePlan(plan, buf, 1024)
    import json
    return json.loads(buf.value.decode())


def i64(values):
    return (ctypes.c_int64 * max(len(values), 1))(*values)


def i32(values):
    return (ctypes.c_int32 * max(len(values), 1))(*[ord(v) if isinstance(v, str) else int(v) for v in values])
