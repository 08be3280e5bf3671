// mg.cpp — cuTENSORMg on a node of MI355X GPUs: single process, one host thread, N devices.
//
// Serves the call sequence of cuTENSORMg/contraction_multi_gpu.cu (create handle :151, block-cyclic
// tensor descriptors :195-217, contraction descriptor / find / workspace / plan :228-250,
// cutensorMgContraction :328-332).  Built entirely on the public single-GPU ABI (include/cutensor.h),
// HIP and RCCL.
//
// Algorithm ("shard the largest free mode, gather the rest"):
//   1. The largest free mode p of C is cut into one contiguous shard per handle device; shards are
//      further cut at p's block boundaries so that every piece lies inside one block.
//   2. Each device gathers the grid cells it needs into its workspace, laid out [cell][cell buffer]:
//      all cells of the operand that does not carry p (an all-gather), the cells of the other
//      operand (and of C when beta != 0) whose p-coordinate it owns.  Cells are whole, contiguous
//      device buffers, so the exchange is plain contiguous transfers: RCCL send/recv pairs inside
//      one ncclGroup over xGMI when every handle device is distinct, device-local copies otherwise.
//   3. The gathered [cell][w, lb] image *is* a tensor: mode i of extent E_i becomes three modes
//      (w_i within a block, c_i grid coordinate, lb_i local block) with strides
//      (elementStride_i, cellElems * cellStride_i, blockStride_i).  The local contraction is one
//      cutensorContract per piece on these views — the GETT engine needs no redistribution kernel.
//   4. Every piece of C is copied from the staging image into the owning cell's buffer by an
//      identity cutensorPermute on strided views (peer stores over xGMI when the owner is remote).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <set>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cutensor.h>
#include <cutensorMg.h>

namespace {

struct MgTensor {
    uint32_t n = 0;
    std::vector<int64_t> extent, elemStride, blockSize, blockStride;
    std::vector<int32_t> deviceCount;
    std::vector<int64_t> localBlocks;   // blocks of each mode stored per cell
    std::vector<int64_t> cellStride;    // linear cell index = sum c_i * cellStride_i
    std::vector<int32_t> devices;       // owner of each cell
    int64_t cellElems = 0;              // elements spanned by one cell buffer
    int64_t numCells = 1;
    hipDataType dtype = HIP_R_32F;
};

size_t elem_size(hipDataType t) {
    switch (t) {
        case HIP_R_16F: case HIP_R_16BF: return 2;
        case HIP_R_32F: return 4;
        case HIP_R_64F: return 8;
        default: return 0;
    }
}

struct Piece {                // a sub-range of the sharded mode handled by one device
    int dev = 0;              // index into the handle's device list
    int64_t lo = 0, hi = 0;   // global index range [lo, hi)
    cutensorPlan_t plan = nullptr;
    uint64_t planWs = 0;
    int64_t offA = 0, offB = 0, offC = 0;      // element offsets of the views in the staging images
    struct Scatter { int cell; int64_t off; cutensorPlan_t plan; };
    std::vector<Scatter> scatter;              // staging C -> owner cell
};

}  // namespace

struct cutensorMgHandle {
    std::vector<int32_t> devices;
    std::vector<cutensorHandle_t> handles;
    bool distinct = true;
    std::vector<ncclComm_t> comms;   // one per handle device when distinct and > 1
};
struct cutensorMgTensorDescriptor { MgTensor t; };
struct cutensorMgContractionDescriptor {
    MgTensor A, B, C, D;
    std::vector<int32_t> mA, mB, mC;
    cutensorComputeType_t compute;
};
struct cutensorMgContractionFind { cutensorMgAlgo_t algo; };
struct cutensorMgContractionPlan {
    cutensorMgContractionDescriptor desc;
    std::vector<Piece> pieces;
    // per handle device: cells to gather (tensor 0 = A, 1 = B, 2 = C)
    std::vector<std::vector<int>> need[3];
    int64_t stagingBytes[3] = {0, 0, 0};
    std::vector<int64_t> wsBytes;    // required per device
    uint64_t contractionWs = 0;
};

namespace {

class DeviceGuard {   // the caller's current device is restored (contraction_multi_gpu.cu:320-346)
public:
    DeviceGuard() { if (hipGetDevice(&saved_) != hipSuccess) saved_ = -1; }
    ~DeviceGuard() { if (saved_ >= 0) (void)hipSetDevice(saved_); }
private:
    int saved_ = -1;
};

int find_label(const std::vector<int32_t>& v, int32_t l) {
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == l) return (int)i;
    return -1;
}

int device_rank(const cutensorMgHandle* h, int32_t dev) {
    for (size_t i = 0; i < h->devices.size(); ++i)
        if (h->devices[i] == dev) return (int)i;
    return -1;
}

// View of tensor T (staging image layout [cell][cell buffer]) for the local contraction.
// Mode `shard` (index in T, or -1) is restricted to [lo, hi); every other mode i becomes up to three
// sub-modes labelled 3*labelIndex + {0,1,2}.
struct View {
    std::vector<int64_t> extent, stride;
    std::vector<int32_t> modes;
    int64_t offset = 0;
};

View make_view(const MgTensor& t, const std::vector<int32_t>& labels, const std::vector<int32_t>& universe,
               int shard, int64_t lo, int64_t hi, bool stagingLayout) {
    View v;
    for (uint32_t i = 0; i < t.n; ++i) {
        const int li = find_label(universe, labels[i]);
        const int64_t cellStep = stagingLayout ? t.cellElems * t.cellStride[i] : 0;
        if ((int)i == shard) {
            const int64_t b = lo / t.blockSize[i];            // global block of the piece
            const int64_t c = b % t.deviceCount[i], lb = b / t.deviceCount[i];
            v.offset += c * cellStep + lb * t.blockStride[i] + (lo - b * t.blockSize[i]) * t.elemStride[i];
            if (hi - lo > 1) { v.extent.push_back(hi - lo); v.stride.push_back(t.elemStride[i]); v.modes.push_back(3 * li); }
            continue;
        }
        if (t.blockSize[i] > 1) { v.extent.push_back(t.blockSize[i]); v.stride.push_back(t.elemStride[i]); v.modes.push_back(3 * li); }
        if (stagingLayout && t.deviceCount[i] > 1) { v.extent.push_back(t.deviceCount[i]); v.stride.push_back(cellStep); v.modes.push_back(3 * li + 1); }
        if (t.localBlocks[i] > 1) { v.extent.push_back(t.localBlocks[i]); v.stride.push_back(t.blockStride[i]); v.modes.push_back(3 * li + 2); }
    }
    return v;
}

cutensorStatus_t make_desc(cutensorHandle_t h, const View& v, hipDataType type, cutensorTensorDescriptor_t* d) {
    return cutensorCreateTensorDescriptor(h, d, (uint32_t)v.extent.size(), v.extent.data(), v.stride.data(), type, 16);
}

cutensorComputeDescriptor_t compute_desc(cutensorComputeType_t c) {
    switch (c) {
        case CUTENSOR_COMPUTE_16F: return CUTENSOR_COMPUTE_DESC_16F;
        case CUTENSOR_COMPUTE_16BF: return CUTENSOR_COMPUTE_DESC_16BF;
        case CUTENSOR_COMPUTE_64F: return CUTENSOR_COMPUTE_DESC_64F;
        default: return CUTENSOR_COMPUTE_DESC_32F;
    }
}

ncclDataType_t nccl_type(hipDataType t) {
    switch (t) {
        case HIP_R_16F: return ncclFloat16;
        case HIP_R_16BF: return ncclBfloat16;
        case HIP_R_64F: return ncclFloat64;
        default: return ncclFloat32;
    }
}

void destroy_pieces(std::vector<Piece>& pieces) {
    for (Piece& p : pieces) {
        cutensorDestroyPlan(p.plan);
        for (auto& s : p.scatter) cutensorDestroyPlan(s.plan);
    }
    pieces.clear();
}

}  // namespace

extern "C" {

// contraction_multi_gpu.cu:151
cutensorStatus_t cutensorMgCreate(cutensorMgHandle_t* handle, uint32_t numDevices, const int32_t devices[]) {
    if (handle == nullptr || numDevices == 0 || devices == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    DeviceGuard guard;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); count = 0; }
    cutensorMgHandle* h = new (std::nothrow) cutensorMgHandle();
    if (h == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    h->devices.assign(devices, devices + numDevices);
    std::set<int32_t> uniq(h->devices.begin(), h->devices.end());
    h->distinct = uniq.size() == h->devices.size();
    for (int32_t d : h->devices) {
        if (d < 0 || (count > 0 && d >= count)) { delete h; return CUTENSOR_STATUS_INVALID_VALUE; }
        if (count > 0) (void)hipSetDevice(d);
        cutensorHandle_t ch = nullptr;
        if (cutensorCreate(&ch) != CUTENSOR_STATUS_SUCCESS) { delete h; return CUTENSOR_STATUS_ALLOC_FAILED; }
        h->handles.push_back(ch);
        if (count > 0)
            for (int32_t o : uniq)
                if (o != d) { (void)hipDeviceEnablePeerAccess(o, 0); (void)hipGetLastError(); }   // xGMI peer mapping
    }
    if (count > 0 && h->distinct && numDevices > 1) {
        h->comms.resize(numDevices);
        std::vector<int> devs(h->devices.begin(), h->devices.end());
        if (ncclCommInitAll(h->comms.data(), (int)numDevices, devs.data()) != ncclSuccess) h->comms.clear();
    }
    *handle = h;
    return CUTENSOR_STATUS_SUCCESS;
}

// contraction_multi_gpu.cu:383
cutensorStatus_t cutensorMgDestroy(cutensorMgHandle_t handle) {
    if (handle == nullptr) return CUTENSOR_STATUS_SUCCESS;
    for (ncclComm_t c : handle->comms) (void)ncclCommDestroy(c);
    for (cutensorHandle_t h : handle->handles) cutensorDestroy(h);
    delete handle;
    return CUTENSOR_STATUS_SUCCESS;
}

// contraction_multi_gpu.cu:195-197
cutensorStatus_t cutensorMgCreateTensorDescriptor(const cutensorMgHandle_t handle, cutensorMgTensorDescriptor_t* desc,
                                                  uint32_t numModes, const int64_t extent[], const int64_t elementStride[],
                                                  const int64_t blockSize[], const int64_t blockStride[],
                                                  const int32_t deviceCount[], uint32_t numDevices, const int32_t devices[],
                                                  cudaDataType_t type) {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || (numModes > 0 && extent == nullptr) || devices == nullptr || numDevices == 0) return CUTENSOR_STATUS_INVALID_VALUE;
    if (numModes > 20 || elem_size(type) == 0) return CUTENSOR_STATUS_NOT_SUPPORTED;
    cutensorMgTensorDescriptor* d = new (std::nothrow) cutensorMgTensorDescriptor();
    if (d == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    MgTensor& t = d->t;
    t.n = numModes;
    t.dtype = type;
    t.extent.assign(extent, extent + numModes);
    t.blockSize.resize(numModes); t.deviceCount.resize(numModes); t.localBlocks.resize(numModes);
    t.elemStride.resize(numModes); t.blockStride.resize(numModes); t.cellStride.resize(numModes);
    int64_t cells = 1, run = 1;
    for (uint32_t i = 0; i < numModes; ++i) {
        if (extent[i] <= 0) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
        t.blockSize[i] = blockSize ? blockSize[i] : extent[i];
        t.deviceCount[i] = deviceCount ? deviceCount[i] : 1;
        if (t.blockSize[i] <= 0 || t.deviceCount[i] <= 0) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
        // uniform block-cyclic layouts only: extent divisible by blockSize * deviceCount (the sample
        // pads to that, contraction_multi_gpu.cu:256)
        if (extent[i] % (t.blockSize[i] * t.deviceCount[i]) != 0) { delete d; return CUTENSOR_STATUS_NOT_SUPPORTED; }
        t.localBlocks[i] = extent[i] / (t.blockSize[i] * t.deviceCount[i]);
        t.cellStride[i] = cells;
        cells *= t.deviceCount[i];
        t.elemStride[i] = elementStride ? elementStride[i] : run;
        run *= t.blockSize[i];
    }
    int64_t brun = run;   // packed block stride: one whole block, then blocks of mode 0, 1, ...
    for (uint32_t i = 0; i < numModes; ++i) {
        t.blockStride[i] = blockStride ? blockStride[i] : brun;
        brun *= t.localBlocks[i];
    }
    if ((int64_t)numDevices != cells) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
    for (uint32_t i = 0; i < numDevices; ++i)
        if (devices[i] == CUTENSOR_MG_DEVICE_HOST) { delete d; return CUTENSOR_STATUS_NOT_SUPPORTED; }
    t.devices.assign(devices, devices + numDevices);
    t.numCells = cells;
    int64_t span = 1;
    for (uint32_t i = 0; i < numModes; ++i)
        span += (t.blockSize[i] - 1) * t.elemStride[i] + (t.localBlocks[i] - 1) * t.blockStride[i];
    t.cellElems = span;
    *desc = d;
    return CUTENSOR_STATUS_SUCCESS;
}

cutensorStatus_t cutensorMgDestroyTensorDescriptor(cutensorMgTensorDescriptor_t desc) {
    delete desc;
    return CUTENSOR_STATUS_SUCCESS;
}

// contraction_multi_gpu.cu:228-233
cutensorStatus_t cutensorMgCreateContractionDescriptor(const cutensorMgHandle_t handle, cutensorMgContractionDescriptor_t* desc,
                                                       const cutensorMgTensorDescriptor_t descA, const int32_t modesA[],
                                                       const cutensorMgTensorDescriptor_t descB, const int32_t modesB[],
                                                       const cutensorMgTensorDescriptor_t descC, const int32_t modesC[],
                                                       const cutensorMgTensorDescriptor_t descD, const int32_t modesD[],
                                                       cutensorComputeType_t compute) {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || descA == nullptr || descB == nullptr || descC == nullptr || descD == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorMgContractionDescriptor* d = new (std::nothrow) cutensorMgContractionDescriptor();
    if (d == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    d->A = descA->t; d->B = descB->t; d->C = descC->t; d->D = descD->t;
    d->mA.assign(modesA, modesA + d->A.n);
    d->mB.assign(modesB, modesB + d->B.n);
    d->mC.assign(modesC, modesC + d->C.n);
    std::vector<int32_t> mD(modesD, modesD + d->D.n);
    d->compute = compute;
    // D must be distributed exactly like C (the sample passes the same descriptor twice)
    const bool sameCD = mD == d->mC && d->C.extent == d->D.extent && d->C.blockSize == d->D.blockSize &&
                        d->C.deviceCount == d->D.deviceCount && d->C.elemStride == d->D.elemStride &&
                        d->C.blockStride == d->D.blockStride && d->C.devices == d->D.devices;
    if (!sameCD || d->A.dtype != d->B.dtype || d->A.dtype != d->C.dtype) { delete d; return CUTENSOR_STATUS_NOT_SUPPORTED; }
    // a mode shared by two tensors must be blocked identically in both
    auto check = [&](const MgTensor& x, const std::vector<int32_t>& mx, const MgTensor& y, const std::vector<int32_t>& my) {
        for (uint32_t i = 0; i < x.n; ++i) {
            const int j = find_label(my, mx[i]);
            if (j < 0) continue;
            if (x.extent[i] != y.extent[j]) return CUTENSOR_STATUS_INVALID_VALUE;
            if (x.blockSize[i] != y.blockSize[j] || x.deviceCount[i] != y.deviceCount[j]) return CUTENSOR_STATUS_NOT_SUPPORTED;
        }
        return CUTENSOR_STATUS_SUCCESS;
    };
    cutensorStatus_t st = check(d->A, d->mA, d->B, d->mB);
    if (st == CUTENSOR_STATUS_SUCCESS) st = check(d->A, d->mA, d->C, d->mC);
    if (st == CUTENSOR_STATUS_SUCCESS) st = check(d->B, d->mB, d->C, d->mC);
    if (st != CUTENSOR_STATUS_SUCCESS) { delete d; return st; }
    *desc = d;
    return CUTENSOR_STATUS_SUCCESS;
}

cutensorStatus_t cutensorMgDestroyContractionDescriptor(cutensorMgContractionDescriptor_t desc) {
    delete desc;
    return CUTENSOR_STATUS_SUCCESS;
}

// contraction_multi_gpu.cu:236-237
cutensorStatus_t cutensorMgCreateContractionFind(const cutensorMgHandle_t handle, cutensorMgContractionFind_t* find,
                                                 const cutensorMgAlgo_t algo) {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (find == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorMgContractionFind* f = new (std::nothrow) cutensorMgContractionFind();
    if (f == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    f->algo = algo;
    *find = f;
    return CUTENSOR_STATUS_SUCCESS;
}

cutensorStatus_t cutensorMgDestroyContractionFind(cutensorMgContractionFind_t find) {
    delete find;
    return CUTENSOR_STATUS_SUCCESS;
}

static const uint64_t kLocalContractionWs = 256ull << 20;   // split-K scratch offered to every local plan

static void staging_sizes(const cutensorMgContractionDescriptor& d, int64_t out[3]) {
    const size_t es = elem_size(d.A.dtype);
    auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
    out[0] = up(d.A.numCells * d.A.cellElems * (int64_t)es);
    out[1] = up(d.B.numCells * d.B.cellElems * (int64_t)es);
    out[2] = up(d.C.numCells * d.C.cellElems * (int64_t)es);
}

// contraction_multi_gpu.cu:241-242
cutensorStatus_t cutensorMgContractionGetWorkspace(const cutensorMgHandle_t handle, const cutensorMgContractionDescriptor_t desc,
                                                   const cutensorMgContractionFind_t find, cutensorWorksizePreference_t preference,
                                                   int64_t deviceWorkspaceSize[], int64_t* hostWorkspaceSize) {
    (void)find; (void)preference;
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || deviceWorkspaceSize == nullptr || hostWorkspaceSize == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    int64_t s[3];
    staging_sizes(*desc, s);
    for (size_t i = 0; i < handle->devices.size(); ++i)
        deviceWorkspaceSize[i] = s[0] + s[1] + s[2] + (int64_t)kLocalContractionWs;
    *hostWorkspaceSize = 0;
    return CUTENSOR_STATUS_SUCCESS;
}

// contraction_multi_gpu.cu:249-250
cutensorStatus_t cutensorMgCreateContractionPlan(const cutensorMgHandle_t handle, cutensorMgContractionPlan_t* plan,
                                                 const cutensorMgContractionDescriptor_t desc, const cutensorMgContractionFind_t find,
                                                 const int64_t deviceWorkspaceSize[], int64_t hostWorkspaceSize) {
    (void)find; (void)hostWorkspaceSize;
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || desc == nullptr || deviceWorkspaceSize == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    DeviceGuard guard;
    cutensorMgContractionPlan* pl = new (std::nothrow) cutensorMgContractionPlan();
    if (pl == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    pl->desc = *desc;
    const cutensorMgContractionDescriptor& d = pl->desc;
    const int nDev = (int)handle->devices.size();
    staging_sizes(d, pl->stagingBytes);
    const int64_t fixed = pl->stagingBytes[0] + pl->stagingBytes[1] + pl->stagingBytes[2];
    pl->wsBytes.assign(nDev, fixed);
    uint64_t ctrWs = kLocalContractionWs;
    for (int g = 0; g < nDev; ++g) {
        if (deviceWorkspaceSize[g] < fixed) { delete pl; return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE; }
        ctrWs = std::min<uint64_t>(ctrWs, (uint64_t)(deviceWorkspaceSize[g] - fixed));
    }
    pl->contractionWs = ctrWs;

    // ---- label universe and the mode to shard -------------------------------------------------
    std::vector<int32_t> universe;
    for (auto* m : {&d.mA, &d.mB, &d.mC})
        for (int32_t l : *m)
            if (find_label(universe, l) < 0) universe.push_back(l);
    int pC = -1;   // index in C of the largest mode that is free (in exactly one of A / B) — else any C mode
    for (int pass = 0; pass < 2 && pC < 0; ++pass)
        for (uint32_t i = 0; i < d.C.n; ++i) {
            const bool inA = find_label(d.mA, d.mC[i]) >= 0, inB = find_label(d.mB, d.mC[i]) >= 0;
            if (pass == 0 && inA == inB) continue;
            if (pC < 0 || d.C.extent[i] > d.C.extent[pC]) pC = (int)i;
        }
    const int32_t pLabel = pC >= 0 ? d.mC[pC] : 0;
    const int pA = pC >= 0 ? find_label(d.mA, pLabel) : -1;
    const int pB = pC >= 0 ? find_label(d.mB, pLabel) : -1;

    // ---- pieces: one shard per device, cut at block boundaries ---------------------------------
    std::vector<Piece> pieces;
    if (pC < 0) {
        Piece p; p.dev = 0; pieces.push_back(p);
    } else {
        const int64_t E = d.C.extent[pC], bs = d.C.blockSize[pC];
        int64_t per = (E + nDev - 1) / nDev;
        per = (per + 15) / 16 * 16;   // keep shard starts 64-byte aligned for the vector kernels
        for (int g = 0; g < nDev; ++g) {
            int64_t lo = (int64_t)g * per, hi = std::min<int64_t>(E, lo + per);
            while (lo < hi) {
                const int64_t cut = std::min<int64_t>(hi, (lo / bs + 1) * bs);
                Piece p; p.dev = g; p.lo = lo; p.hi = cut;
                pieces.push_back(p);
                lo = cut;
            }
        }
    }

    // ---- which cells each device gathers --------------------------------------------------------
    const MgTensor* T[3] = {&d.A, &d.B, &d.C};
    const int shardIdx[3] = {pA, pB, pC};
    for (int k = 0; k < 3; ++k) pl->need[k].assign(nDev, std::vector<int>());
    for (int g = 0; g < nDev; ++g) {
        std::set<int64_t> coords;   // grid coordinates of mode p touched by device g
        for (const Piece& p : pieces)
            if (p.dev == g && pC >= 0) coords.insert((p.lo / d.C.blockSize[pC]) % d.C.deviceCount[pC]);
        const bool active = std::any_of(pieces.begin(), pieces.end(), [&](const Piece& p) { return p.dev == g; });
        if (!active) continue;
        for (int k = 0; k < 3; ++k) {
            const MgTensor& t = *T[k];
            for (int64_t c = 0; c < t.numCells; ++c) {
                bool want = true;
                if (shardIdx[k] >= 0) {
                    const int64_t coord = (c / t.cellStride[shardIdx[k]]) % t.deviceCount[shardIdx[k]];
                    want = coords.count(coord) > 0;
                }
                if (want) pl->need[k][g].push_back((int)c);
            }
        }
    }

    // ---- local plans ----------------------------------------------------------------------------
    const cutensorComputeDescriptor_t cd = compute_desc(d.compute);
    cutensorStatus_t st = CUTENSOR_STATUS_SUCCESS;
    for (Piece& p : pieces) {
        (void)hipSetDevice(handle->devices[p.dev]);
        (void)hipGetLastError();
        cutensorHandle_t h = handle->handles[p.dev];
        const View vA = make_view(d.A, d.mA, universe, pA, p.lo, p.hi, true);
        const View vB = make_view(d.B, d.mB, universe, pB, p.lo, p.hi, true);
        const View vC = make_view(d.C, d.mC, universe, pC, p.lo, p.hi, true);
        p.offA = vA.offset; p.offB = vB.offset; p.offC = vC.offset;
        cutensorTensorDescriptor_t dA = nullptr, dB = nullptr, dC = nullptr;
        cutensorOperationDescriptor_t op = nullptr;
        cutensorPlanPreference_t pref = nullptr;
        st = make_desc(h, vA, d.A.dtype, &dA);
        if (st == CUTENSOR_STATUS_SUCCESS) st = make_desc(h, vB, d.B.dtype, &dB);
        if (st == CUTENSOR_STATUS_SUCCESS) st = make_desc(h, vC, d.C.dtype, &dC);
        if (st == CUTENSOR_STATUS_SUCCESS)
            st = cutensorCreateContraction(h, &op, dA, vA.modes.data(), CUTENSOR_OP_IDENTITY, dB, vB.modes.data(), CUTENSOR_OP_IDENTITY,
                                           dC, vC.modes.data(), CUTENSOR_OP_IDENTITY, dC, vC.modes.data(), cd);
        if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlanPreference(h, &pref, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE);
        if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlan(h, &p.plan, op, pref, pl->contractionWs);
        if (st == CUTENSOR_STATUS_SUCCESS)
            st = cutensorPlanGetAttribute(h, p.plan, CUTENSOR_PLAN_REQUIRED_WORKSPACE, &p.planWs, sizeof(p.planWs));
        cutensorDestroyOperationDescriptor(op);
        cutensorDestroyPlanPreference(pref);
        cutensorDestroyTensorDescriptor(dA);
        cutensorDestroyTensorDescriptor(dB);
        cutensorDestroyTensorDescriptor(dC);
        if (st != CUTENSOR_STATUS_SUCCESS) break;
        // scatter plans: the piece's region inside every cell of C it touches (cell-relative strides)
        const View cellView = make_view(d.C, d.mC, universe, pC, p.lo, p.hi, false);
        const int64_t coord = pC >= 0 ? (p.lo / d.C.blockSize[pC]) % d.C.deviceCount[pC] : 0;
        for (int64_t c = 0; c < d.C.numCells && st == CUTENSOR_STATUS_SUCCESS; ++c) {
            if (pC >= 0 && (c / d.C.cellStride[pC]) % d.C.deviceCount[pC] != coord) continue;
            cutensorTensorDescriptor_t ds = nullptr;
            cutensorOperationDescriptor_t po = nullptr;
            Piece::Scatter s{(int)c, cellView.offset, nullptr};
            st = make_desc(h, cellView, d.C.dtype, &ds);
            if (st == CUTENSOR_STATUS_SUCCESS)
                st = cutensorCreatePermutation(h, &po, ds, cellView.modes.data(), CUTENSOR_OP_IDENTITY, ds, cellView.modes.data(), cd);
            if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlan(h, &s.plan, po, nullptr, 0);
            cutensorDestroyOperationDescriptor(po);
            cutensorDestroyTensorDescriptor(ds);
            if (st == CUTENSOR_STATUS_SUCCESS) p.scatter.push_back(s);
        }
        if (st != CUTENSOR_STATUS_SUCCESS) break;
    }
    if (st != CUTENSOR_STATUS_SUCCESS) {
        destroy_pieces(pieces);
        delete pl;
        return st;
    }
    pl->pieces.swap(pieces);
    *plan = pl;
    return CUTENSOR_STATUS_SUCCESS;
}

cutensorStatus_t cutensorMgDestroyContractionPlan(cutensorMgContractionPlan_t plan) {
    if (plan == nullptr) return CUTENSOR_STATUS_SUCCESS;
    destroy_pieces(plan->pieces);
    delete plan;
    return CUTENSOR_STATUS_SUCCESS;
}

// contraction_multi_gpu.cu:328-332
cutensorStatus_t cutensorMgContraction(const cutensorMgHandle_t handle, const cutensorMgContractionPlan_t plan,
                                       const void* alpha, const void* A[], const void* B[], const void* beta,
                                       const void* C[], void* D[], void* workspaceDevice[], void* workspaceHost,
                                       cudaStream_t streams[]) {
    (void)workspaceHost;
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || alpha == nullptr || beta == nullptr || A == nullptr || B == nullptr || D == nullptr ||
        workspaceDevice == nullptr || streams == nullptr)
        return CUTENSOR_STATUS_INVALID_VALUE;
    DeviceGuard guard;
    const cutensorMgContractionDescriptor& d = plan->desc;
    const int nDev = (int)handle->devices.size();
    const size_t es = elem_size(d.A.dtype);
    const bool f64 = d.A.dtype == HIP_R_64F || d.compute == CUTENSOR_COMPUTE_64F;
    const double b = f64 ? *static_cast<const double*>(beta) : (double)*static_cast<const float*>(beta);
    if (b != 0.0 && C == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    const MgTensor* T[3] = {&d.A, &d.B, &d.C};
    const void* const* src[3] = {A, B, C};

    auto staging = [&](int g, int k) -> char* {
        char* base = static_cast<char*>(workspaceDevice[g]);
        for (int j = 0; j < k; ++j) base += plan->stagingBytes[j];
        return base;
    };

    // ---- 1. gather ------------------------------------------------------------------------------
    const bool useRccl = !handle->comms.empty();
    bool grouped = false;
    for (int g = 0; g < nDev; ++g) {
        for (int k = 0; k < 3; ++k) {
            if (k == 2 && b == 0.0) continue;
            const MgTensor& t = *T[k];
            const size_t cellBytes = (size_t)t.cellElems * es;
            for (int c : plan->need[k][g]) {
                char* dst = staging(g, k) + (size_t)c * cellBytes;
                const int32_t owner = t.devices[c];
                const int ownerRank = device_rank(handle, owner);
                if (owner == handle->devices[g]) {
                    (void)hipSetDevice(owner);
                    if (hipMemcpyAsync(dst, src[k][c], cellBytes, hipMemcpyDeviceToDevice, streams[g]) != hipSuccess)
                        return CUTENSOR_STATUS_EXECUTION_FAILED;
                } else if (useRccl && ownerRank >= 0) {
                    if (!grouped) { (void)ncclGroupStart(); grouped = true; }
                    (void)hipSetDevice(owner);
                    if (ncclSend(src[k][c], (size_t)t.cellElems, nccl_type(t.dtype), g, handle->comms[ownerRank], streams[ownerRank]) != ncclSuccess)
                        return CUTENSOR_STATUS_EXECUTION_FAILED;
                    (void)hipSetDevice(handle->devices[g]);
                    if (ncclRecv(dst, (size_t)t.cellElems, nccl_type(t.dtype), ownerRank, handle->comms[g], streams[g]) != ncclSuccess)
                        return CUTENSOR_STATUS_EXECUTION_FAILED;
                } else {
                    (void)hipSetDevice(handle->devices[g]);
                    if (hipMemcpyPeerAsync(dst, handle->devices[g], src[k][c], owner, cellBytes, streams[g]) != hipSuccess)
                        return CUTENSOR_STATUS_EXECUTION_FAILED;
                }
            }
        }
    }
    if (grouped && ncclGroupEnd() != ncclSuccess) return CUTENSOR_STATUS_EXECUTION_FAILED;

    // ---- 2. local contractions, 3. scatter -------------------------------------------------------
    for (const Piece& p : plan->pieces) {
        const int g = p.dev;
        (void)hipSetDevice(handle->devices[g]);
        char* sA = staging(g, 0);
        char* sB = staging(g, 1);
        char* sC = staging(g, 2);
        char* ws = staging(g, 3);
        cutensorStatus_t st = cutensorContract(handle->handles[g], p.plan, alpha, sA + p.offA * (int64_t)es, sB + p.offB * (int64_t)es,
                                               beta, sC + p.offC * (int64_t)es, sC + p.offC * (int64_t)es, ws, plan->contractionWs, streams[g]);
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
        const float onef = 1.f;
        const double oned = 1.0;
        const void* one = f64 ? static_cast<const void*>(&oned) : static_cast<const void*>(&onef);
        const size_t cellBytes = (size_t)d.C.cellElems * es;
        for (const Piece::Scatter& s : p.scatter) {
            const char* from = sC + (size_t)s.cell * cellBytes + s.off * (int64_t)es;
            char* to = static_cast<char*>(D[s.cell]) + s.off * (int64_t)es;
            st = cutensorPermute(handle->handles[g], s.plan, one, from, to, streams[g]);
            if (st != CUTENSOR_STATUS_SUCCESS) return st;
        }
    }
    return CUTENSOR_STATUS_SUCCESS;
}

}  // extern "C"
