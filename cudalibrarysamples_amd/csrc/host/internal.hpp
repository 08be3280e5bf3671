// internal.hpp — objects behind the opaque cuTENSOR handles and the planner interfaces.
#pragma once
#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <cutensor.h>

#include "api_guard.hpp"

// Measurement-only environment switches (CUTENSOR_AMD_FORCE, _XCD_BALANCE, _KORDER, _ABLATION, _PARTIAL_STORE, _H16_SPLITK,
// _H16_TRANSPOSE_T1, ...) are read in research builds only (make RESEARCH=1): the production library answers "not set".  What stays
// live in production are the switches the test-suite and the tools drive behaviour with: CUTENSOR_LOG_LEVEL, CUTENSOR_AMD_GEN, _PEEL, _NT,
// _FUSED_FOLD, _H16_WAVES, _H16P_GRID and the CUTENSORMG_AMD_* / CUTENSORMP_AMD_* ones.
#include <cstdlib>
inline const char* ctamd_research_env(const char* name) {
#if defined(CTAMD_RESEARCH_KERNELS)
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
#include "../kernels/launch.h"
#include "../kernels/params.h"

// The ABI names these structs only as opaque pointer targets (cutensor/types.h).
struct cutensorComputeDescriptor {
    int      id;          // index into the exported constants
    uint32_t legacyBits;  // cutensorComputeType_t value (Mg entry points)
};

struct cutensorTensorDescriptor {
    uint32_t             numModes = 0;
    std::vector<int64_t> extent;
    std::vector<int64_t> stride;
    hipDataType          dtype = HIP_R_32F;
    uint32_t             alignment = 0;
    int64_t numElementsSpanned() const;  // 1 + sum (extent-1)*stride
};

enum class OpKind : int { Contraction = 0, Reduction = 1, Permutation = 2, ElementwiseBinary = 3, ElementwiseTrinary = 4,
                           ContractionTrinary = 5, BlockSparseContraction = 6 };

struct TensorUse {
    cutensorTensorDescriptor desc;
    std::vector<int32_t>     modes;
    cutensorOperator_t       op = CUTENSOR_OP_IDENTITY;
    bool                     present = false;
};

namespace ctamd { struct BlockSparseOp; struct BlockSparsePlan; }

struct cutensorOperationDescriptor {
    OpKind      kind;
    std::shared_ptr<ctamd::BlockSparseOp> bs;        // block-sparse contraction (host/blocksparse.cpp)
    TensorUse   A, B, C, D;
    TensorUse   E;                                   // output of a trinary contraction (D is its beta source)
    cutensorOperator_t opReduce = CUTENSOR_OP_ADD;   // reduction operator / binary combiner (opAC, opABC)
    cutensorOperator_t opAB = CUTENSOR_OP_ADD;       // element-wise trinary: first combiner
    const cutensorComputeDescriptor* compute = nullptr;
    hipDataType scalarType = HIP_R_32F;
    int32_t     tag = 0;
    // trinary contraction E = alpha A B C + beta D (contraction_trinary.cu:191-198) as two pairwise contractions:
    // sub[0]: T = X * Y (packed intermediate in the workspace), sub[1]: E = alpha T * Z + beta D
    std::vector<cutensorOperationDescriptor> sub;
    int         triOrder[3] = {0, 1, 2};   // operand indices (0 = A, 1 = B, 2 = C) playing X, Y, Z
    uint64_t    tBytes = 0;
    // CUTENSOR_OPERATION_DESCRIPTOR_PADDING_{LEFT,RIGHT,VALUE} of a permutation (elementwise_permute_padding.cu:178-195)
    std::vector<int32_t> padLeft, padRight;
    double      padValue = 0.0;
    double      flops = 0.0;
    double      movedBytes = 0.0;
};

struct cutensorPlanPreference {
    cutensorAlgo_t         algo = CUTENSOR_ALGO_DEFAULT;
    cutensorJitMode_t      jit = CUTENSOR_JIT_MODE_NONE;
    cutensorAutotuneMode_t autotune = CUTENSOR_AUTOTUNE_MODE_NONE;
    cutensorCacheMode_t    cacheMode = CUTENSOR_CACHE_MODE_PEDANTIC;
    int32_t                incrementalCount = 4;
    int32_t                kernelRank = 0;
    int32_t                operandsStreamed = 0;   // CUTENSOR_AMD_PLAN_PREFERENCE_OPERANDS_STREAMED (engine extension)
};

namespace ctamd {

// ---- contraction planning --------------------------------------------------------------------
struct CanonMode {
    int32_t label;
    int64_t extent;
    int64_t sA = 0, sB = 0, sC = 0, sD = 0;   // element strides (kernel-A / kernel-B / C / D)
};

struct ContractionView {
    std::vector<CanonMode> L, M, N, K;   // fused, fastest first
    bool     swapped = false;            // kernel-A is the user's B
    int      layA = LAY_S, layB = LAY_S; // best layout each operand admits
    hipDataType dtype = HIP_R_32F;
    uint64_t totL = 1, totM = 1, totN = 1, totK = 1;
    bool     wide = false;               // a group has more than kMaxGroupModes unfusable modes (or >= 2^31 elements):
                                         // only the mode-table kernel (gett_wide_kernel) can run it
    bool     lanesA = false, lanesB = false;   // fp32: the operand has 16-byte lanes in the strict sense (pointer, extent of the stride-1
                                         // mode and every other stride multiples of 16 bytes) — what the register-staged kernels
                                         // (gett_f32.hip) need; the LDS-DMA ring kernels take layA / layB as they are (round 6)
    uint32_t alignA = 0, alignB = 0;     // descriptor alignment (bytes) of kernel-A / kernel-B
    uint32_t alignD = 0;                 // ... of the output
};

// One executable choice for a contraction.
struct ContractionChoice {
    int      kernel = -1;      // index into the family's kernel table; -1 = simple kernel
    int      family = 0;       // 0 = gett_f32_kernels() (fp32 data), 1 = gett_h16_kernels() (bf16 / fp16 data, aligned shapes),
                               // 2 = gett_gen_kernels() (general MFMA family: any 16-bit shape, fp64, complex)
    uint32_t splitK = 1;
    uint32_t kPerSlice = 0;
    uint64_t workspace = 0;
    double   estimateUs = 0.0;
    // Strip plan (16-bit family, pick_h16_choice): `kernel` covers only the interior [0, mInt) x [0, nInt) — whole tiles of its own size —
    // and ONE launch of table entry `stripKernel` (the 64 x 64 tile) covers the two edge strips, rows [mInt, M) x all columns and
    // rows [0, mInt) x columns [nInt, N).  4100^3 on 256 x 256 tiles is 17 x 17 = 289 tiles, two rounds on 256 CUs for 33 tiles that
    // hold four live rows or columns each; as 16 x 16 + strips it is one round plus ~20 us.  stripKernel < 0: none.
    int      stripKernel = -1;
    uint32_t mInt = 0, nInt = 0;
};

cutensorStatus_t build_contraction_view(const cutensorOperationDescriptor& op, ContractionView& v,
                                        std::string* why);
// Ranked candidate list (best first) under a workspace limit.
std::vector<ContractionChoice> rank_contraction_choices(const ContractionView& v, uint64_t wsLimit,
                                                        int numCUs, bool operandsStreamed = false);
bool pick_h16_choice(const ContractionView& v, uint64_t wsLimit, int numCUs, ContractionChoice& c);
// general MFMA family (kernels/gett_gen.inc): false only for fp32 data and for views the tiled kernels cannot describe
bool pick_gen_choice(const ContractionView& v, uint64_t wsLimit, int numCUs, ContractionChoice& c);
// 16-bit family: the default kernel variant first, then the other variants of the same tile / split (the candidates
// CUTENSOR_ALGO_DEFAULT_PATIENT and incremental autotuning measure)
std::vector<ContractionChoice> rank_h16_choices(const ContractionView& v, uint64_t wsLimit, int numCUs);
void fill_gett_params(const ContractionView& v, const ContractionChoice& c, GettParams& p,
                      SplitKReduceParams& r);

// ---- element-wise / reduction planning -------------------------------------------------------
struct EwPlan {
    int        variant = EW_GENERIC;
    Ew2DParams p{};              // pointers / scalars are filled at launch
    bool       usesC = false;
    bool       usesX = false;    // second permuted operand (trinary, both operands permuted)
};
// cutensorElementwiseTrinaryExecute: D = opABC(opAB(alpha A, beta B), gamma C) as one or two passes of the
// element-wise kernels (plan_elementwise_trinary)
struct EwTrinaryPlan {
    bool   twoPass = false;      // pass 1: D = s1 * perm(X1);  last pass: D = opAC(opAB(delta * E, s2 * perm(X2)), gamma * perm(C))
    bool   swapAB = false;       // the operand that already has D's layout plays E (single pass): true -> E = B, X2 = A
    bool   bothPermuted = false; // single pass with two LDS tiles: A through the tile path, B as the second tile operand
    EwPlan first;                // permutation X1 -> D (two-pass form only)
    EwPlan last;
};
struct ReducePlan {
    int          variant = RED_GENERIC;
    ReduceParams p{};
    uint64_t     workspace = 0;
    bool         isPermutation = false;   // no reduced modes: executed by the element-wise family
    EwPlan       perm;
};

// allowWide = false: never the tiled kernels of 8- / 16-byte elements (the trinary planner: they take no E / X operand)
cutensorStatus_t plan_elementwise(const cutensorOperationDescriptor& op, EwPlan& plan, std::string* why, bool allowWide = true);
cutensorStatus_t plan_elementwise_trinary(const cutensorOperationDescriptor& op, EwTrinaryPlan& plan, std::string* why);
cutensorStatus_t plan_reduction(const cutensorOperationDescriptor& op, uint64_t wsLimit, int numCUs,
                                ReducePlan& plan, std::string* why);

cutensorStatus_t blocksparse_estimate(cutensorHandle_t handle, const cutensorOperationDescriptor& desc, uint64_t* ws);
cutensorStatus_t blocksparse_plan(cutensorHandle_t handle, const cutensorOperationDescriptor& desc, uint64_t wsLimit, cutensorPlan* pl);

FastDiv make_fastdiv(uint32_t d);
size_t  dtype_size(hipDataType t);

}  // namespace ctamd

// Incremental-autotuning trial plans are timed by cutensorContract for their first few executions only; the counter lives in
// the (otherwise immutable) plan, copyable so that plans stay copy-constructible (the plan memo clones prototypes).
struct TrialCounter {
    mutable std::atomic<int> timed{0};
    TrialCounter() = default;
    TrialCounter(const TrialCounter& o) : timed(o.timed.load(std::memory_order_relaxed)) {}
    TrialCounter& operator=(const TrialCounter& o) { timed.store(o.timed.load(std::memory_order_relaxed), std::memory_order_relaxed); return *this; }
};

// A contraction whose mode groups are too wide for the tiled kernels' argument block, peeled: the modes listed here are walked
// by the host (one launch of the inner, tiled plan per index combination, operands offset by index x stride); a contracted
// peeled mode accumulates into D, a free one writes a region of its own.
struct PeelMode {
    int64_t extent = 1;
    int64_t sA = 0, sB = 0, sC = 0, sD = 0;   // element strides of the peeled label in the user's A, B, C, D (0: not carried)
    bool    contracted = false;
};

struct cutensorPlan {
    cutensorPlan() = default;
    cutensorPlan(const cutensorPlan&) = default;   // valid only for plans that own nothing (sub1/sub2/wide.modes null): the memo's clones
    ~cutensorPlan();
    ctamd::WideParams wide{};        // mode-table contraction (view.wide): .modes is device memory owned by this plan,
    std::vector<ctamd::WideMode> wideTab;   // uploaded from this host copy by the first cutensorContract
    // trinary contraction: the two pairwise plans, the intermediate's size and which operand plays which role
    std::shared_ptr<ctamd::BlockSparsePlan> bsp;     // block-sparse contraction: dense plans + block-pair task list
    cutensorPlan* sub1 = nullptr;                    // (also: the inner plan of a peeled contraction, choice.kernel == -3)
    cutensorPlan* sub2 = nullptr;
    std::vector<PeelMode> peel;
    // contraction with a mode that one input alone carries (choice.kernel == -4, api.cpp split_lone_modes): reductions of A / B over
    // those modes into packed temporaries at the head of the workspace (nullptr: the operand is used as it is); sub1 = the contraction
    cutensorPlan* loneA = nullptr;
    cutensorPlan* loneB = nullptr;
    uint64_t    loneBytesA = 0, loneBytesB = 0;
    uint64_t    tBytes = 0;
    int         triOrder[3] = {0, 1, 2};
    OpKind      kind;
    hipDataType dtype = HIP_R_32F;
    hipDataType scalarType = HIP_R_32F;
    uint64_t    requiredWorkspace = 0;
    uint32_t    alignA = 0, alignB = 0, alignC = 0, alignD = 0;  // pointer alignment the plan relies on
    // contraction
    ctamd::ContractionView     view;
    ctamd::ContractionChoice   choice;
    ctamd::GettParams          gett{};
    ctamd::SplitKReduceParams  skr{};
    bool                       accumulate64 = false;
    bool                       fusedFold = false;     // split-K partials are folded inside the GETT launch
    std::string                tuneKey;               // non-empty: an incremental-autotuning trial, timed by cutensorContract
    TrialCounter               trial;                 // executions of this trial plan timed so far (at most kTrialTimedRuns)
    static constexpr int       kTrialTimedRuns = 3;
    // element-wise / reduction
    ctamd::EwPlan     ew;
    ctamd::EwTrinaryPlan ew3;
    // padded permutation: the output buffer holds extents + padLeft + padRight per mode; it is filled with the
    // padding value, then the permutation writes the interior
    uint64_t          padFillElems = 0;    // 0 = no padding
    int64_t           padOffsetElems = 0;  // element offset of the interior's origin
    double            padValue = 0.0;
    uint32_t          alignB3 = 0;     // element-wise trinary: alignment of B
    ctamd::ReducePlan red;
};

struct PlanCacheEntry {
    std::string key;
    int         kernel;
    uint32_t    splitK;
};

// Plan memo: the einsum.cu flow creates descriptors and a plan inside every call (einsum.cu:264-329) with the plan cache on
// (:443-445), so a repeated problem must cost a lookup, not a planning pass.  The key is a fixed-size POD built straight from
// the operation descriptor + preference + workspace limit (no strings, no allocation); the value is a finished prototype plan
// that a hit clones.  Problems with more than kMaxModes mode slots in total are simply not memoised.
struct PlanMemoKey {
    static constexpr int kMaxModes = 40;          // sum of the mode counts of A, B, C, D
    uint64_t wsLimit = 0;
    int32_t  algo = 0, kernelRank = 0, autotune = 0, incrementalCount = 0;
    uint32_t alignment[4] = {0, 0, 0, 0};
    uint8_t  kind = 0, dtype = 0, compute = 0, scalarType = 0;
    uint8_t  n[4] = {0, 0, 0, 0};
    uint8_t  op[4] = {0, 0, 0, 0};                 // opA, opB, opC, opReduce
    uint8_t  present = 0, operandsStreamed = 0, pad_[2] = {0, 0};
    uint32_t used = 0;                             // int64 words of data[] in use
    uint32_t pad2_ = 0;                            // (no implicit padding anywhere in the head: it is hashed and compared bytewise)
    int64_t  data[3 * kMaxModes];                  // per tensor: modes, extents, strides
    uint64_t hash() const;
    bool operator==(const PlanMemoKey& o) const;
};
struct PlanMemoEntry {
    PlanMemoKey key;
    std::shared_ptr<const cutensorPlan> proto;
    uint64_t stamp = 0;                            // last use, for LRU eviction
};

struct cutensorHandle {
    int device = 0;
    bool haveDevice = false;            // a GPU was visible at cutensorCreate (false: descriptors and plans only)
    int numCUs = 256;
    int clockKHz = 2400000;
    std::mutex mtx;
    uint32_t planCacheCapacity = 64;    // a fresh handle can read a plan-cache file before any resize (contraction_plan_cache.cu:132-152)
    std::map<std::string, PlanCacheEntry> planCache;   // problem signature -> tuned choice (what the cache FILE holds)
    std::unordered_map<uint64_t, PlanMemoEntry> planMemo;   // hashed POD key -> finished prototype plan (at most planCacheCapacity)
    uint64_t memoClock = 0;
    std::atomic<uint64_t> memoHits{0}, memoMisses{0};
    std::atomic<int> pendingCount{0};                  // == pending.size(), readable without the lock
    int logLevel = 0;
    // {arrivals, departures} counter pairs for in-launch split-K folds, 64 B apart, zeroed once; a launch
    // draws the next slot round-robin and its last workgroup re-arms it (gett_f32_stream.hip)
    // relative sustained shader clock of the 8 XCDs under the streaming GETT kernel (measured once per handle by
    // calibrate_xcd_split); empty = not measured, all-equal = measurement failed / disabled
    std::vector<double> xcdSpeed;
    uint32_t* syncPool = nullptr;
    uint32_t  syncNext = 0;
    static constexpr uint32_t kSyncSlots = 256;
    // diagnostics, per handle (ctamdSetSplitKFold / ctamdSetTimingBuffer / ctamdProfileBegin): a second handle — another
    // framework thread's — never sees them
    std::atomic<bool> skipFold{false};
    std::atomic<unsigned long long*> timingBuffer{nullptr};
    struct KernelProfile {
        std::atomic<bool> enabled{false};
        std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
        std::mutex mtx;
    } prof;
    // incremental autotuning (CUTENSOR_AUTOTUNE_MODE_INCREMENTAL, contraction_plan_cache.cu:215-237): per problem, the
    // candidates tried so far with their measured time; the best one is what the plan cache holds
    struct TuneState {
        int      next = 0;                 // next candidate rank to try
        int      bestKernel = -1;
        uint32_t bestSplitK = 1;
        float    bestMs = 1e30f;
    };
    std::map<std::string, TuneState> tuning;
    struct PendingMeasurement { std::string key; int kernel; uint32_t splitK; hipEvent_t e0, e1; };
    std::vector<PendingMeasurement> pending;   // cutensorContract calls of tuning plans whose events were not read yet
};
