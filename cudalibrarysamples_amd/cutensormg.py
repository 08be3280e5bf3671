"""ctypes mirror of include/cutensorMg.h (call sites: cuTENSORMg/contraction_multi_gpu.cu:151-383)."""
import ctypes
import os

from . import cutensor as ct   # loads libcutensor.so with RTLD_GLOBAL first

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib_hooks" if os.environ.get("CTAMD_LIB_FLAVOUR") == "hooks" else "lib", "libcutensorMg.so")
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

lib.ctamdMgDescribePlan.argtypes = [_vp, ctypes.c_char_p, ctypes.c_size_t]
lib.ctamdMgDescribePlan.restype = ctypes.c_int
lib.ctamdMgDescribeKBoxes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_char_p, ctypes.c_size_t]
lib.ctamdMgDescribeKBoxes.restype = ctypes.c_int



class HostView(ctypes.Structure):
    """ctamdMgHostView: one operand view of a local contraction (extents, element strides, mode labels 8 * label + digit)."""
    _fields_ = [("n", ctypes.c_int32), ("extent", _i64p), ("stride", _i64p), ("modes", _i32p)]


HOST_CONTRACT_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(HostView), ctypes.c_void_p, ctypes.POINTER(HostView),
                                    ctypes.c_void_p, ctypes.POINTER(HostView), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double)
HAVE_REPLAY = hasattr(lib, "ctamdMgReplayOnHost")      # a test entry point: the hooks flavour only
if HAVE_REPLAY:
    lib.ctamdMgReplayOnHost.argtypes = [_vp, ctypes.c_double, _vpp, _vpp, ctypes.c_double, _vpp, _vpp, HOST_CONTRACT_FN, ctypes.c_void_p,
                                        ctypes.c_char_p, ctypes.c_size_t]
    lib.ctamdMgReplayOnHost.restype = ctypes.c_int

check = ct.check
i64, i32 = ct.i64, ct.i32


def replay_on_host(plan, alpha, A, B, beta, C, D, contract):
    """ctamdMgReplayOnHost: executes the plan over HOST cell buffers (lists of addresses) — transfers as memcpy into NaN-filled host
    staging images, local contractions through `contract(dtype, viewA, ptrA, viewB, ptrB, viewC, ptrC, ptrD, alpha, beta) -> 0`
    (views as (extent, stride, modes) tuples), scatters as strided copies, and checks that every piece has waited for the events
    that carry the cells it reads.  Returns (rc, message).  A test hook: the caller supplies the reference contraction."""
    def cb(user, dtype, va, pa, vb, pb, vc, pc, pd, al, be):
        try:
            views = [(list(v.contents.extent[:v.contents.n]), list(v.contents.stride[:v.contents.n]), list(v.contents.modes[:v.contents.n]))
                     for v in (va, vb, vc)]
            return int(contract(dtype, views[0], pa, views[1], pb, views[2], pc, pd, al, be))
        except Exception:   # noqa: BLE001  (an exception must not cross the C frame)
            import traceback
            traceback.print_exc()
            return -1
    fn = HOST_CONTRACT_FN(cb)
    err = ctypes.create_string_buffer(512)
    rc = lib.ctamdMgReplayOnHost(plan, float(alpha), ptr_array(A), ptr_array(B), float(beta), ptr_array(C) if C is not None else None,
                                 ptr_array(D), fn, None, err, len(err))
    return rc, err.value.decode()


def kboxes(extent, block_size, digits):
    """ctamdMgDescribeKBoxes -> list of {"wHi", "digits": [[lo, hi], ...]}: the boxes that tile the valid part of a ragged
    contracted mode's padded index space (mg.cpp kbox_list)."""
    import json
    buf = ctypes.create_string_buffer(1 << 14)
    f = (ctypes.c_int64 * max(len(digits), 1))(*digits)
    r = lib.ctamdMgDescribeKBoxes(extent, block_size, len(digits), f, buf, len(buf))
    if r < 0:
        raise ValueError("ctamdMgDescribeKBoxes: invalid arguments")
    return json.loads(buf.value.decode())


def describe_plan(plan):
    """ctamdMgDescribePlan -> dict: sharded / ordering mode labels, pieces in execution order, cell transfers."""
    import json
    n = 1 << 16
    while True:
        buf = ctypes.create_string_buffer(n)
        r = lib.ctamdMgDescribePlan(plan, buf, n)
        if r >= 0:
            return json.loads(buf.value.decode())
        if r == -1:
            raise ValueError("ctamdMgDescribePlan: invalid arguments")
        n = -r + 16


class Contraction:
    """RAII bundle of the cuTENSORMg call sequence of contraction_multi_gpu.cu:151-250 for C[mc] = A[ma] * B[mb]:
    handle over `devices`, three block-cyclic descriptors (extent / blockSize / deviceCount per mode label, cells owned
    by `cell_devices[k]`, default: handle devices repeated cyclically as the sample's fillUp() does), contraction
    descriptor, find, workspace query and plan."""

    def __init__(self, devices, modes, extent, block, dcount, cell_devices=None, dtype=0, compute=COMPUTE_32F):
        self.devices = list(devices)
        self.modes = modes
        self.h = ctypes.c_void_p()
        check(cutensorMgCreate(ctypes.byref(self.h), len(devices), i32(self.devices)))
        self.descs, self.cells = [], []
        for k, m in enumerate(modes):
            ext = [extent[c] for c in m]
            bs = [block[k].get(c, extent[c]) for c in m]
            dc = [dcount[k].get(c, 1) for c in m]
            ncell = 1
            for x in dc:
                ncell *= x
            owners = list(cell_devices[k]) if cell_devices is not None else [self.devices[i % len(self.devices)] for i in range(ncell)]
            d = ctypes.c_void_p()
            check(cutensorMgCreateTensorDescriptor(self.h, ctypes.byref(d), len(m), i64(ext), None, i64(bs), None, i32(dc), ncell,
                                                   i32(owners), dtype))
            self.descs.append(d)
            self.cells.append(dict(ext=ext, bs=bs, dc=dc, owners=owners, ncell=ncell))
        lab = [i32([ord(c) for c in m]) for m in modes]
        self.cd = ctypes.c_void_p()
        check(cutensorMgCreateContractionDescriptor(self.h, ctypes.byref(self.cd), self.descs[0], lab[0], self.descs[1], lab[1],
                                                    self.descs[2], lab[2], self.descs[2], lab[2], compute))
        self.find = ctypes.c_void_p()
        check(cutensorMgCreateContractionFind(self.h, ctypes.byref(self.find), ALGO_DEFAULT))
        n = len(devices)
        self.ws_sizes = (ctypes.c_int64 * n)()
        self.host_size = ctypes.c_int64(0)
        check(cutensorMgContractionGetWorkspace(self.h, self.cd, self.find, 2, self.ws_sizes, ctypes.byref(self.host_size)))
        self.plan = ctypes.c_void_p()
        check(cutensorMgCreateContractionPlan(self.h, ctypes.byref(self.plan), self.cd, self.find, self.ws_sizes, self.host_size.value))

    def describe(self):
        return describe_plan(self.plan)

    def run(self, alpha, A, B, beta, C, D, workspaces, streams, scalar=ctypes.c_float):
        a, b = scalar(alpha), scalar(beta)
        return cutensorMgContraction(self.h, self.plan, ctypes.byref(a), ptr_array(A), ptr_array(B), ctypes.byref(b), ptr_array(C),
                                     ptr_array(D), ptr_array(workspaces), None, ptr_array(streams))

    def close(self):
        if self.plan:
            check(cutensorMgDestroyContractionPlan(self.plan))
            check(cutensorMgDestroyContractionFind(self.find))
            check(cutensorMgDestroyContractionDescriptor(self.cd))
            for d in self.descs:
                check(cutensorMgDestroyTensorDescriptor(d))
            check(cutensorMgDestroy(self.h))
            self.plan = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def ptr_array(values):
    return (ctypes.c_void_p * max(len(values), 1))(*values)
