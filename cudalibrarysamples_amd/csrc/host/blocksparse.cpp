// blocksparse.cpp — block-sparse tensor contraction on top of the dense engine.
//
// Reference call sites: cuTENSOR/blocksparse.cu:102-107 (cutensorCreateBlockSparseTensorDescriptor: a tensor is a
// set of dense blocks; every mode is cut into sections, a block is addressed by one section index per mode and
// lives behind its own device pointer, packed column-major unless strides are given), :177-182
// (cutensorCreateBlockSparseContraction), :191-197 (workspace estimate "is exact", plan), :206-209
// (cutensorBlockSparseContract with arrays of block pointers).
//
// D = alpha * A * B + beta * C over blocks: for every pair (block of A, block of B) whose section indices agree on
// the modes they share, the dense contraction of the two blocks is accumulated into the output block addressed by
// their free modes' sections (pairs whose output block is not stored are structural zeros and skipped).  Each
// distinct block-shape triple gets one dense plan of the GETT engine; the first contribution to an output block
// carries the caller's beta, the following ones beta = 1; output blocks without any contribution are scaled by
// beta (or cleared).  Work is issued block pair by block pair on the caller's stream.
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "internal.hpp"

using namespace ctamd;

struct cutensorBlockSparseTensorDescriptor {
    uint32_t numModes = 0;
    uint64_t nnz = 0;
    std::vector<std::vector<int64_t>> sections;   // [mode][section] extent
    std::vector<int32_t> coords;                  // [block][mode]
    std::vector<int64_t> strides;                 // [block][mode], empty = packed
    hipDataType dtype = HIP_R_64F;
};

namespace ctamd {

struct BlockSparseOp {
    cutensorBlockSparseTensorDescriptor A, B, C, D;
    std::vector<int32_t> mA, mB, mC, mD;
    cutensorComputeDescriptor_t compute = nullptr;
};

struct BlockSparseTask {
    int64_t a = -1, b = -1, d = -1;   // block indices (a = b = -1: no contribution, D = beta * C)
    int     plan = -1;                // index into plans
    bool    first = true;             // first contribution to block d
};

struct BlockSparsePlan {
    std::vector<BlockSparseTask> tasks;
    std::vector<cutensorPlan_t>  plans;      // dense contraction plans (one per distinct shape triple)
    std::vector<cutensorPlan_t>  scalePlans; // identity permutations for blocks without contributions (beta != 0)
    std::vector<int>             scaleOf;    // per task (a < 0): index into scalePlans
    std::vector<uint64_t>        blockElems; // elements of every D block
    ~BlockSparsePlan() {
        for (cutensorPlan_t p : plans) cutensorDestroyPlan(p);
        for (cutensorPlan_t p : scalePlans) cutensorDestroyPlan(p);
    }
};

static void block_shape(const cutensorBlockSparseTensorDescriptor& T, uint64_t blk, std::vector<int64_t>& ext, std::vector<int64_t>& str) {
    ext.resize(T.numModes);
    str.resize(T.numModes);
    int64_t run = 1;
    for (uint32_t m = 0; m < T.numModes; ++m) {
        ext[m] = T.sections[m][(size_t)T.coords[blk * T.numModes + m]];
        str[m] = T.strides.empty() ? run : T.strides[blk * T.numModes + m];
        run *= ext[m];
    }
}

// Enumerates the block pairs and builds (or only sizes, when handle plans are not wanted) the dense plans.
static cutensorStatus_t build_blocksparse(cutensorHandle_t handle, const BlockSparseOp& op, uint64_t wsLimit, bool makePlans,
                                          BlockSparsePlan* out, uint64_t* workspace) {
    auto find = [](const std::vector<int32_t>& v, int32_t l) { for (size_t i = 0; i < v.size(); ++i) if (v[i] == l) return (int)i; return -1; };
    if (op.mC != op.mD) return CUTENSOR_STATUS_NOT_SUPPORTED;
    // index of the D block with given per-mode sections
    std::map<std::vector<int32_t>, int64_t> dIndex, cIndex;
    for (uint64_t i = 0; i < op.D.nnz; ++i)
        dIndex[std::vector<int32_t>(op.D.coords.begin() + i * op.D.numModes, op.D.coords.begin() + (i + 1) * op.D.numModes)] = (int64_t)i;
    for (uint64_t i = 0; i < op.C.nnz; ++i)
        cIndex[std::vector<int32_t>(op.C.coords.begin() + i * op.C.numModes, op.C.coords.begin() + (i + 1) * op.C.numModes)] = (int64_t)i;
    for (const auto& kv : dIndex)   // the engine reads beta * C from the block of C at the same position
        if (cIndex.find(kv.first) == cIndex.end() || cIndex[kv.first] != kv.second) return CUTENSOR_STATUS_NOT_SUPPORTED;

    std::vector<char> touched(op.D.nnz, 0);
    std::map<std::string, int> planOf;
    uint64_t ws = 0;
    for (uint64_t a = 0; a < op.A.nnz; ++a)
        for (uint64_t b = 0; b < op.B.nnz; ++b) {
            bool match = true;
            std::vector<int32_t> dc(op.D.numModes, -1);
            for (uint32_t i = 0; i < op.A.numModes && match; ++i) {
                const int32_t l = op.mA[i], sa = op.A.coords[a * op.A.numModes + i];
                const int ib = find(op.mB, l), id = find(op.mD, l);
                if (ib >= 0 && op.B.coords[b * op.B.numModes + ib] != sa) match = false;
                if (id >= 0) dc[id] = sa;
            }
            for (uint32_t i = 0; i < op.B.numModes && match; ++i) {
                const int id = find(op.mD, op.mB[i]);
                if (id >= 0) {
                    const int32_t sb = op.B.coords[b * op.B.numModes + i];
                    if (dc[id] >= 0 && dc[id] != sb) match = false;
                    dc[id] = sb;
                }
            }
            if (!match) continue;
            auto it = dIndex.find(dc);
            if (it == dIndex.end()) continue;   // structural zero of the output
            const int64_t d = it->second;
            BlockSparseTask t;
            t.a = (int64_t)a; t.b = (int64_t)b; t.d = d;
            t.first = !touched[d];
            touched[d] = 1;
            // dense plan for this shape triple
            std::vector<int64_t> eA, sA, eB, sB, eD, sD;
            block_shape(op.A, a, eA, sA);
            block_shape(op.B, b, eB, sB);
            block_shape(op.D, (uint64_t)d, eD, sD);
            std::string key;
            for (auto* v : {&eA, &sA, &eB, &sB, &eD, &sD}) { for (int64_t x : *v) key += std::to_string(x) + ","; key += "|"; }
            auto pit = planOf.find(key);
            if (pit == planOf.end()) {
                cutensorTensorDescriptor_t dA = nullptr, dB = nullptr, dD = nullptr;
                cutensorOperationDescriptor_t od = nullptr;
                const uint32_t al = (uint32_t)dtype_size(op.A.dtype);
                cutensorStatus_t st = cutensorCreateTensorDescriptor(handle, &dA, op.A.numModes, eA.data(), sA.data(), op.A.dtype, al);
                if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreateTensorDescriptor(handle, &dB, op.B.numModes, eB.data(), sB.data(), op.B.dtype, al);
                if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreateTensorDescriptor(handle, &dD, op.D.numModes, eD.data(), sD.data(), op.D.dtype, al);
                if (st == CUTENSOR_STATUS_SUCCESS)
                    st = cutensorCreateContraction(handle, &od, dA, op.mA.data(), CUTENSOR_OP_IDENTITY, dB, op.mB.data(), CUTENSOR_OP_IDENTITY,
                                                   dD, op.mD.data(), CUTENSOR_OP_IDENTITY, dD, op.mD.data(), op.compute);
                uint64_t w = 0;
                if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorEstimateWorkspaceSize(handle, od, nullptr, CUTENSOR_WORKSPACE_DEFAULT, &w);
                if (st == CUTENSOR_STATUS_SUCCESS) {
                    ws = std::max(ws, w);
                    if (makePlans) {
                        cutensorPlan_t pl = nullptr;
                        st = cutensorCreatePlan(handle, &pl, od, nullptr, std::min(w, wsLimit));
                        if (st == CUTENSOR_STATUS_SUCCESS) out->plans.push_back(pl);
                    }
                }
                cutensorDestroyOperationDescriptor(od);
                cutensorDestroyTensorDescriptor(dA);
                cutensorDestroyTensorDescriptor(dB);
                cutensorDestroyTensorDescriptor(dD);
                if (st != CUTENSOR_STATUS_SUCCESS) return st;
                pit = planOf.emplace(key, (int)planOf.size()).first;
            }
            t.plan = pit->second;
            if (makePlans) out->tasks.push_back(t);
        }
    if (makePlans) {
        std::map<std::string, int> scaleOf;
        out->blockElems.resize(op.D.nnz);
        for (uint64_t d = 0; d < op.D.nnz; ++d) {
            std::vector<int64_t> eD, sD;
            block_shape(op.D, d, eD, sD);
            uint64_t n = 1;
            for (int64_t e : eD) n *= (uint64_t)e;
            out->blockElems[d] = n;
            if (touched[d]) continue;
            BlockSparseTask t;
            t.d = (int64_t)d;
            std::string key;
            for (auto* v : {&eD, &sD}) { for (int64_t x : *v) key += std::to_string(x) + ","; key += "|"; }
            auto sit = scaleOf.find(key);
            if (sit == scaleOf.end()) {
                cutensorTensorDescriptor_t dD = nullptr;
                cutensorOperationDescriptor_t od = nullptr;
                cutensorPlan_t pl = nullptr;
                cutensorStatus_t st = cutensorCreateTensorDescriptor(handle, &dD, op.D.numModes, eD.data(), sD.data(), op.D.dtype,
                                                                     (uint32_t)dtype_size(op.D.dtype));
                if (st == CUTENSOR_STATUS_SUCCESS)
                    st = cutensorCreatePermutation(handle, &od, dD, op.mD.data(), CUTENSOR_OP_IDENTITY, dD, op.mD.data(), op.compute);
                if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlan(handle, &pl, od, nullptr, 0);
                cutensorDestroyOperationDescriptor(od);
                cutensorDestroyTensorDescriptor(dD);
                if (st != CUTENSOR_STATUS_SUCCESS) return st;
                out->scalePlans.push_back(pl);
                sit = scaleOf.emplace(key, (int)scaleOf.size()).first;
            }
            t.plan = sit->second;
            out->tasks.push_back(t);
        }
    }
    if (workspace) *workspace = ws;
    return CUTENSOR_STATUS_SUCCESS;
}

cutensorStatus_t blocksparse_estimate(cutensorHandle_t handle, const cutensorOperationDescriptor& desc, uint64_t* ws) {
    if (!desc.bs) return CUTENSOR_STATUS_INVALID_VALUE;
    return build_blocksparse(handle, *desc.bs, ~0ull, false, nullptr, ws);
}

cutensorStatus_t blocksparse_plan(cutensorHandle_t handle, const cutensorOperationDescriptor& desc, uint64_t wsLimit, cutensorPlan* pl) {
    if (!desc.bs) return CUTENSOR_STATUS_INVALID_VALUE;
    auto bp = std::make_shared<BlockSparsePlan>();
    uint64_t ws = 0;
    cutensorStatus_t st = build_blocksparse(handle, *desc.bs, wsLimit, true, bp.get(), &ws);
    if (st != CUTENSOR_STATUS_SUCCESS) return st;
    uint64_t need = 0;
    for (cutensorPlan_t p : bp->plans) need = std::max(need, p->requiredWorkspace);
    pl->bsp = bp;
    pl->dtype = desc.bs->D.dtype;
    pl->requiredWorkspace = need;
    return CUTENSOR_STATUS_SUCCESS;
}

}  // namespace ctamd

extern "C" {

// blocksparse.cu:102-107
cutensorStatus_t cutensorCreateBlockSparseTensorDescriptor(cutensorHandle_t handle, cutensorBlockSparseTensorDescriptor_t* desc,
                                                           const uint32_t numModes, const uint64_t numNonZeroBlocks,
                                                           const uint32_t numSectionsPerMode[], const int64_t extent[],
                                                           const int32_t nonZeroCoordinates[], const int64_t stride[],
                                                           cudaDataType_t dataType) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || numModes == 0 || numSectionsPerMode == nullptr || extent == nullptr || (numNonZeroBlocks && nonZeroCoordinates == nullptr))
        return CUTENSOR_STATUS_INVALID_VALUE;
    if (dtype_size(dataType) == 0) return CUTENSOR_STATUS_NOT_SUPPORTED;
    auto* d = new (std::nothrow) cutensorBlockSparseTensorDescriptor();
    if (d == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    d->numModes = numModes;
    d->nnz = numNonZeroBlocks;
    d->dtype = dataType;
    size_t off = 0;
    for (uint32_t m = 0; m < numModes; ++m) {
        d->sections.emplace_back(extent + off, extent + off + numSectionsPerMode[m]);
        for (int64_t e : d->sections.back()) if (e <= 0) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
        off += numSectionsPerMode[m];
    }
    d->coords.assign(nonZeroCoordinates, nonZeroCoordinates + numNonZeroBlocks * numModes);
    for (uint64_t i = 0; i < numNonZeroBlocks; ++i)
        for (uint32_t m = 0; m < numModes; ++m) {
            const int32_t c = d->coords[i * numModes + m];
            if (c < 0 || (uint32_t)c >= numSectionsPerMode[m]) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
        }
    if (stride != nullptr) d->strides.assign(stride, stride + numNonZeroBlocks * numModes);
    *desc = d;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

cutensorStatus_t cutensorDestroyBlockSparseTensorDescriptor(cutensorBlockSparseTensorDescriptor_t desc) try {
    delete desc;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// blocksparse.cu:177-182
cutensorStatus_t cutensorCreateBlockSparseContraction(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                                      const cutensorBlockSparseTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                                      const cutensorBlockSparseTensorDescriptor_t descB, const int32_t modeB[], cutensorOperator_t opB,
                                                      const cutensorBlockSparseTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                                      const cutensorBlockSparseTensorDescriptor_t descD, const int32_t modeD[],
                                                      const cutensorComputeDescriptor_t descCompute) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || descA == nullptr || descB == nullptr || descC == nullptr || descD == nullptr || descCompute == nullptr)
        return CUTENSOR_STATUS_INVALID_VALUE;
    if (opA != CUTENSOR_OP_IDENTITY || opB != CUTENSOR_OP_IDENTITY || opC != CUTENSOR_OP_IDENTITY) return CUTENSOR_STATUS_NOT_SUPPORTED;
    if (descA->dtype != descB->dtype || descA->dtype != descC->dtype || descA->dtype != descD->dtype) return CUTENSOR_STATUS_NOT_SUPPORTED;
    // C is read through D's block descriptors and untouched output blocks are cleared as packed blocks: C must be laid out
    // exactly like D (the sample passes one descriptor for both, blocksparse.cu:177-182) and D's blocks must be packed
    if (descC->sections != descD->sections || descC->strides != descD->strides || descC->coords != descD->coords || !descD->strides.empty())
        return CUTENSOR_STATUS_NOT_SUPPORTED;
    auto bs = std::make_shared<BlockSparseOp>();
    bs->A = *descA; bs->B = *descB; bs->C = *descC; bs->D = *descD;
    bs->mA.assign(modeA, modeA + descA->numModes);
    bs->mB.assign(modeB, modeB + descB->numModes);
    bs->mC.assign(modeC, modeC + descC->numModes);
    bs->mD.assign(modeD, modeD + descD->numModes);
    bs->compute = descCompute;
    auto* op = new (std::nothrow) cutensorOperationDescriptor();
    if (op == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    op->kind = OpKind::BlockSparseContraction;
    op->compute = descCompute;
    op->bs = bs;
    // real data only: the per-block scalars below are real (complex block-sparse contractions are not on the path)
    if (descA->dtype == HIP_C_32F || descA->dtype == HIP_C_64F) { delete op; return CUTENSOR_STATUS_NOT_SUPPORTED; }
    op->scalarType = (descA->dtype == HIP_R_64F || descCompute->id == 5) ? HIP_R_64F : HIP_R_32F;
    uint64_t ws = 0;
    const cutensorStatus_t st = build_blocksparse(handle, *bs, ~0ull, false, nullptr, &ws);   // validates shapes / modes
    if (st != CUTENSOR_STATUS_SUCCESS) { delete op; return st; }
    *desc = op;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// blocksparse.cu:206-209
cutensorStatus_t cutensorBlockSparseContract(const cutensorHandle_t handle, const cutensorPlan_t plan, const void* alpha,
                                             const void* const A[], const void* const B[], const void* beta,
                                             const void* const C[], void* const D[], void* workspace, uint64_t workspaceSize,
                                             cudaStream_t stream) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || plan->kind != OpKind::BlockSparseContraction || !plan->bsp) return CUTENSOR_STATUS_INVALID_VALUE;
    if (alpha == nullptr || beta == nullptr || A == nullptr || B == nullptr || D == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (plan->requiredWorkspace > 0 && (workspace == nullptr || workspaceSize < plan->requiredWorkspace)) return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE;
    const bool f64 = plan->scalarType == HIP_R_64F;
    const double b = f64 ? *static_cast<const double*>(beta) : (double)*static_cast<const float*>(beta);
    if (b != 0.0 && C == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    const double one64 = 1.0;
    const float one32 = 1.f;
    const void* one = f64 ? static_cast<const void*>(&one64) : static_cast<const void*>(&one32);
    const BlockSparsePlan& bp = *plan->bsp;
    for (const BlockSparseTask& t : bp.tasks) {
        cutensorStatus_t st;
        if (t.a >= 0) {
            // first contribution: beta * C[d]; later ones accumulate onto D[d]
            st = cutensorContract(handle, bp.plans[(size_t)t.plan], alpha, A[t.a], B[t.b], t.first ? beta : one,
                                  t.first ? (b != 0.0 ? C[t.d] : D[t.d]) : D[t.d], D[t.d], workspace, workspaceSize, stream);
        } else if (b == 0.0) {
            st = launch_fill(D[t.d], bp.blockElems[(size_t)t.d], (int)plan->dtype, 0.0, stream) == hipSuccess ? CUTENSOR_STATUS_SUCCESS
                                                                                                               : CUTENSOR_STATUS_EXECUTION_FAILED;
        } else {
            st = cutensorPermute(handle, bp.scalePlans[(size_t)t.plan], beta, C[t.d], D[t.d], stream);   // D = beta * C
        }
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
    }
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

}  // extern "C"
