// api.cpp — the exported cuTENSOR C ABI (include/cutensor.h) on top of the planners and the
// gfx950 kernels.  Each entry point cites the reference call site it serves.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <sstream>

#include <hip/hip_runtime.h>

#include "internal.hpp"

using namespace ctamd;

// ---- compute descriptor constants (einsum.cu:39,46,53; contraction.cu:40) ----------------------
static const cutensorComputeDescriptor kCompute[6] = {
    {0, CUTENSOR_COMPUTE_16F}, {1, CUTENSOR_COMPUTE_16BF}, {2, CUTENSOR_COMPUTE_TF32},
    {3, CUTENSOR_COMPUTE_3XTF32}, {4, CUTENSOR_COMPUTE_32F}, {5, CUTENSOR_COMPUTE_64F}};
extern "C" {
const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_16F    = &kCompute[0];
const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_16BF   = &kCompute[1];
const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_TF32   = &kCompute[2];
const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_3XTF32 = &kCompute[3];
const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_32F    = &kCompute[4];
const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_64F    = &kCompute[5];
}

// grid of the persistent 16-bit kernel in workgroups (0 = one per CU): CUTENSOR_AMD_H16P_GRID, a test hook — the kernel objects are
// shared by both library flavours, so the switch is read here and handed over as data (gett_h16p.hip, launch_h16w4p)
extern "C" int ctamd_h16p_grid_cap;
extern "C" int ctamd_h16p_stagger;
int ctamd_h16p_stagger = [] { const char* e = CTAMD_HOOK_ENV("CUTENSOR_AMD_H16P_STAGGER"); return e ? std::atoi(e) : -1; }();   // -1: the launcher's own step
int ctamd_h16p_grid_cap = [] { const char* e = CTAMD_HOOK_ENV("CUTENSOR_AMD_H16P_GRID"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 0; }();

namespace {

// launches of cutensorContract by kernel kind since the library was loaded (ctamdLaunchCounts): 0 = gett_simple_kernel (scalar FMA
// fallback), 1 = gett_wide_kernel (mode table), 2 = fp32 MFMA families, 3 = aligned 16-bit MFMA family, 4 = general MFMA family
std::atomic<uint64_t> g_launchCounts[5];
std::atomic<int> g_lastH16Kernel{-1};   // table entry of the last launch of the aligned 16-bit family (ctamdLastH16Kernel: which twin ran)

bool valid_compute(cutensorComputeDescriptor_t c) { return c >= &kCompute[0] && c <= &kCompute[5]; }

int log_level() {
    static int lvl = [] {
        const char* e = std::getenv("CUTENSOR_LOG_LEVEL");   // contraction_jit.cu:142 hints at this knob
        return e ? std::atoi(e) : 0;
    }();
    return lvl;
}
#define CT_LOG(...) do { if (log_level() > 0) { std::fprintf(stderr, "[cutensor-amd] " __VA_ARGS__); std::fputc('\n', stderr); } } while (0)

bool supported_dtype(hipDataType t) {
    // complex data: contractions (general MFMA family / mode-table kernel), reductions, permutations and the binary element-wise
    // form (ADD / MUL); the trinary element-wise planner answers NOT_SUPPORTED for it
    return t == HIP_R_32F || t == HIP_R_64F || t == HIP_R_16F || t == HIP_R_16BF || t == HIP_C_32F || t == HIP_C_64F;
}

// scalar type of alpha/beta for a data type + compute descriptor (einsum.cu:40,47,54;
// torch/einsum.cc:39): fp64 data -> fp64 scalars, everything else -> fp32 scalars.
hipDataType scalar_type_for(hipDataType data, const cutensorComputeDescriptor* c) {
    if (data == HIP_C_32F || data == HIP_C_64F) return data;   // contraction_jit.cu:205: complex scalars for complex data
    if (data == HIP_R_64F || (c && c->id == 5)) return HIP_R_64F;
    return HIP_R_32F;
}

cutensorStatus_t fill_use(TensorUse& u, const cutensorTensorDescriptor_t d, const int32_t* modes,
                          cutensorOperator_t op) {
    if (d == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (d->numModes > 0 && modes == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    u.desc = *d;
    u.modes.assign(modes, modes + d->numModes);
    u.op = op;
    u.present = true;
    return CUTENSOR_STATUS_SUCCESS;
}

double num_elements(const cutensorTensorDescriptor& d) {
    double n = 1.0;
    for (int64_t e : d.extent) n *= (double)e;
    return n;
}

std::string problem_key(const cutensorOperationDescriptor& op) {
    std::ostringstream ss;
    ss << (int)op.kind << ':' << (int)op.A.desc.dtype << ':' << (op.compute ? op.compute->id : -1);
    auto put = [&](const TensorUse& u) {
        ss << '|';
        if (!u.present) return;
        for (size_t i = 0; i < u.modes.size(); ++i)
            ss << u.modes[i] << ',' << u.desc.extent[i] << ',' << u.desc.stride[i] << ';';
        ss << 'a' << u.desc.alignment;
    };
    put(op.A); put(op.B); put(op.C); put(op.D);
    return ss.str();
}

// POD memo key (internal.hpp): false when the problem is outside what the memo holds
bool build_memo_key(const cutensorOperationDescriptor& op, const cutensorPlanPreference& pr, uint64_t wsLimit, PlanMemoKey& k) {
    if (op.kind != OpKind::Contraction && op.kind != OpKind::Reduction && op.kind != OpKind::Permutation &&
        op.kind != OpKind::ElementwiseBinary) return false;
    if (!op.padLeft.empty() || !op.padRight.empty()) return false;
    const TensorUse* u[4] = {&op.A, &op.B, &op.C, &op.D};
    size_t total = 0;
    for (const TensorUse* t : u) if (t->present) total += t->modes.size();
    if (total > (size_t)PlanMemoKey::kMaxModes) return false;
    k.wsLimit = wsLimit;
    k.algo = (int32_t)pr.algo; k.kernelRank = pr.kernelRank; k.autotune = (int32_t)pr.autotune;
    k.incrementalCount = pr.autotune == CUTENSOR_AUTOTUNE_MODE_INCREMENTAL ? pr.incrementalCount : 0;
    k.operandsStreamed = (uint8_t)(pr.operandsStreamed != 0);
    k.kind = (uint8_t)op.kind; k.dtype = (uint8_t)op.A.desc.dtype; k.compute = (uint8_t)(op.compute ? op.compute->id : 255);
    k.scalarType = (uint8_t)op.scalarType;
    k.op[0] = (uint8_t)op.A.op; k.op[1] = (uint8_t)op.B.op; k.op[2] = (uint8_t)op.C.op; k.op[3] = (uint8_t)op.opReduce;
    k.present = 0;
    uint32_t w = 0;
    for (int i = 0; i < 4; ++i) {
        const TensorUse& t = *u[i];
        k.n[i] = 0; k.alignment[i] = 0;
        if (!t.present) continue;
        k.present |= (uint8_t)(1u << i);
        const size_t n = t.modes.size();
        k.n[i] = (uint8_t)n; k.alignment[i] = t.desc.alignment;
        for (size_t j = 0; j < n; ++j) k.data[w++] = t.modes[j];
        std::memcpy(&k.data[w], t.desc.extent.data(), n * sizeof(int64_t)); w += (uint32_t)n;
        std::memcpy(&k.data[w], t.desc.stride.data(), n * sizeof(int64_t)); w += (uint32_t)n;
    }
    k.used = w;
    return true;
}

}  // namespace

uint64_t PlanMemoKey::hash() const {
    // the fixed head (everything before data[]) and the used words, 8 bytes at a time through a multiply-xorshift mix
    auto mix = [](uint64_t h, uint64_t v) { h ^= v; h *= 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); };
    uint64_t h = 0xCBF29CE484222325ull;
    const size_t headWords = offsetof(PlanMemoKey, data) / 8;
    const uint64_t* p = reinterpret_cast<const uint64_t*>(this);
    for (size_t i = 0; i < headWords; ++i) h = mix(h, p[i]);
    for (uint32_t i = 0; i < used; ++i) h = mix(h, (uint64_t)data[i]);
    return h;
}
bool PlanMemoKey::operator==(const PlanMemoKey& o) const {
    return used == o.used && std::memcmp(this, &o, offsetof(PlanMemoKey, data)) == 0 &&
           std::memcmp(data, o.data, (size_t)used * sizeof(int64_t)) == 0;
}

namespace {

// any experiment knob that changes what cutensorCreatePlan decides: the memo stands aside while one is set
bool plan_env_override() {
    return ctamd_research_env("CUTENSOR_AMD_FORCE") || ctamd_research_env("CUTENSOR_AMD_XCD_BALANCE") || CTAMD_HOOK_ENV("CUTENSOR_AMD_FUSED_FOLD") ||
           ctamd_research_env("CUTENSOR_AMD_H16_TRANSPOSE_T1") || CTAMD_HOOK_ENV("CUTENSOR_AMD_NT") || CTAMD_HOOK_ENV("CUTENSOR_AMD_H16_WAVES") || ctamd_research_env("CUTENSOR_AMD_H16_SPLITK") ||
           ctamd_research_env("CUTENSOR_AMD_KORDER") || ctamd_research_env("CUTENSOR_AMD_ABLATION") || CTAMD_HOOK_ENV("CUTENSOR_AMD_PEEL") || CTAMD_HOOK_ENV("CUTENSOR_AMD_GEN") ||
           CTAMD_HOOK_ENV("CUTENSOR_AMD_REPACK") || CTAMD_HOOK_ENV("CUTENSOR_AMD_EW_ANY");
}

double scalar_as_double(const void* s, hipDataType t) {   // real part for complex scalar types
    if (s == nullptr) return 0.0;
    return (t == HIP_R_64F || t == HIP_C_64F) ? *static_cast<const double*>(s) : (double)*static_cast<const float*>(s);
}
double scalar_imag(const void* s, hipDataType t) {
    if (s == nullptr) return 0.0;
    if (t == HIP_C_64F) return static_cast<const double*>(s)[1];
    if (t == HIP_C_32F) return (double)static_cast<const float*>(s)[1];
    return 0.0;
}

bool misaligned(const void* p, uint32_t a) { return a > 1 && (reinterpret_cast<uintptr_t>(p) % a) != 0; }

}  // namespace

// Incremental autotuning (contraction_plan_cache.cu:215-237): cutensorContract on a trial plan brackets its launches with
// an event pair; the pairs are read here — at the next plan creation, before the cache is written, at handle destruction —
// never inside cutensorContract, which stays asynchronous.  The fastest candidate seen so far is what the cache holds.
static void resolve_pending_measurements(cutensorHandle* handle) {
    std::vector<cutensorHandle::PendingMeasurement> todo;
    {
        std::lock_guard<std::mutex> g(handle->mtx);
        todo.swap(handle->pending);
        handle->pendingCount.store(0, std::memory_order_relaxed);
    }
    for (auto& m : todo) {
        float ms = 0.f;
        const bool ok = hipEventSynchronize(m.e1) == hipSuccess && hipEventElapsedTime(&ms, m.e0, m.e1) == hipSuccess;
        (void)hipEventDestroy(m.e0);
        (void)hipEventDestroy(m.e1);
        if (!ok) { (void)hipGetLastError(); continue; }
        std::lock_guard<std::mutex> g(handle->mtx);
        cutensorHandle::TuneState& t = handle->tuning[m.key];
        CT_LOG("incremental autotune: kernel %d splitK %u -> %.3f us (best so far %.3f us)", m.kernel, m.splitK, ms * 1e3, t.bestMs * 1e3);
        if (ms < t.bestMs) {
            t.bestMs = ms; t.bestKernel = m.kernel; t.bestSplitK = m.splitK;
            if (handle->planCache.count(m.key) || handle->planCache.size() < handle->planCacheCapacity) {
                handle->planCache[m.key] = PlanCacheEntry{m.key, m.kernel, m.splitK};
                handle->planMemo.clear();   // prototypes built from the previous best are stale
            }
        }
    }
}

// A finished plan that owns nothing becomes the prototype later plans of the same problem are cloned from; the least
// recently used prototype makes room when the cache is full (capacity = cutensorHandleResizePlanCache's numEntries).
// A plan holds device memory, a trial's timing state or a block-sparse task list of its own: nothing a copy may share
static bool plan_is_plain(const cutensorPlan& pl) {
    return pl.tuneKey.empty() && pl.wide.modes == nullptr && pl.wideTab.empty() && !pl.bsp;
}
// Copy of a plan; the plans of a two-step contraction (choice.kernel == -4: an operand reduced over its lone modes or copied into a packed
// temporary first, then the inner contraction) are copied with it — every one of them plain (round 6: such plans are memoised too; the
// planner prices up to eight copy combinations for them, 100-240 us per cutensorCreatePlan, and einsum.cu plans inside every call)
static cutensorPlan* clone_plan(const cutensorPlan& src) {
    cutensorPlan* c = new (std::nothrow) cutensorPlan(src);
    if (c == nullptr) return nullptr;
    c->sub1 = c->sub2 = c->loneA = c->loneB = nullptr;
    const cutensorPlan* from[4] = {src.sub1, src.sub2, src.loneA, src.loneB};
    cutensorPlan** to[4] = {&c->sub1, &c->sub2, &c->loneA, &c->loneB};
    for (int i = 0; i < 4; ++i)
        if (from[i] != nullptr && (*to[i] = clone_plan(*from[i])) == nullptr) { delete c; return nullptr; }
    return c;
}
static bool plan_is_prototype(const cutensorPlan& pl) {
    if (!plan_is_plain(pl)) return false;
    if (pl.sub1 == nullptr && pl.sub2 == nullptr && pl.loneA == nullptr && pl.loneB == nullptr) return true;
    if (pl.kind != OpKind::Contraction || pl.choice.kernel != -4 || pl.sub2 != nullptr || pl.sub1 == nullptr) return false;   // (peeled / trinary plans: not memoised)
    for (const cutensorPlan* q : {static_cast<const cutensorPlan*>(pl.sub1), static_cast<const cutensorPlan*>(pl.loneA), static_cast<const cutensorPlan*>(pl.loneB)})
        if (q != nullptr && (!plan_is_plain(*q) || q->sub1 || q->sub2 || q->loneA || q->loneB)) return false;
    return true;
}
static void memo_insert(cutensorHandle* h, const PlanMemoKey& key, uint64_t hash, const cutensorPlan& pl) {
    if (!plan_is_prototype(pl)) return;   // plans that own something are not prototypes
    std::shared_ptr<const cutensorPlan> proto(clone_plan(pl));
    if (!proto) return;
    std::lock_guard<std::mutex> g(h->mtx);
    if (h->planCacheCapacity == 0) return;
    if (h->planMemo.size() >= h->planCacheCapacity && h->planMemo.find(hash) == h->planMemo.end()) {
        auto victim = h->planMemo.begin();
        for (auto it = h->planMemo.begin(); it != h->planMemo.end(); ++it)
            if (it->second.stamp < victim->second.stamp) victim = it;
        h->planMemo.erase(victim);
    }
    PlanMemoEntry& e = h->planMemo[hash];
    e.key = key; e.proto = std::move(proto); e.stamp = ++h->memoClock;
}

// Peeling of a "wide" contraction (a group with more than kMaxGroupModes unfusable modes would run on the functional
// mode-table kernel, measured ~100x slower than the tiled ones): labels of the oversized groups are taken out of the problem,
// smallest extent first, until the rest fits the tiled kernels — at most kMaxPeelLaunches index combinations, each one launch of
// the same inner plan on offset operands.  Returns false when the problem is not wide for that reason or would take more launches.
static constexpr int64_t kMaxPeelLaunches = 64;
static bool peel_wide_contraction(const cutensorOperationDescriptor& desc, cutensorOperationDescriptor& inner, std::vector<PeelMode>& peel) {
    if (desc.kind != OpKind::Contraction) return false;
    inner = desc;
    peel.clear();
    // 16-bit data: a peeled CONTRACTED mode would accumulate through D, i.e. round every partial sum to the 16-bit type — the
    // kernels' contract is fp32 accumulation and ONE rounding.  Only free / batch modes are peeled for these types; an oversized K
    // group stays with the mode-table kernel (which accumulates in full precision).
    const bool h16 = dtype_size(desc.A.desc.dtype) == 2;
    int64_t launches = 1;
    auto idx = [](const TensorUse& t, int32_t l) { for (size_t i = 0; i < t.modes.size(); ++i) if (t.modes[i] == l) return (int)i; return -1; };
    for (int round = 0; round < 16; ++round) {
        ContractionView v;
        if (build_contraction_view(inner, v, nullptr) != CUTENSOR_STATUS_SUCCESS) return false;
        if (!v.wide) return !peel.empty();
        const std::vector<CanonMode>* gs[4] = {&v.L, &v.M, &v.N, &v.K};
        int32_t bestLabel = 0;
        int64_t bestExtent = 0;
        for (int g = 0; g < 4; ++g) {
            if ((int)gs[g]->size() <= kMaxGroupModes) continue;
            if (g == 3 && h16) return false;
            for (const CanonMode& m : *gs[g]) {
                // the label's own extent (a canonical mode may be a fused run; peeling its first label shortens the run)
                const int ia = idx(inner.A, m.label), ib = idx(inner.B, m.label), ic = idx(inner.C, m.label);
                const int64_t e = ia >= 0 ? inner.A.desc.extent[(size_t)ia] : ib >= 0 ? inner.B.desc.extent[(size_t)ib] : ic >= 0 ? inner.C.desc.extent[(size_t)ic] : 0;
                if (e < 2) continue;
                if (bestExtent == 0 || e < bestExtent) { bestLabel = m.label; bestExtent = e; }
            }
        }
        if (bestExtent == 0) return false;                       // wide for another reason (>= 2^31 elements in a group)
        launches *= bestExtent;
        if (launches > kMaxPeelLaunches) return false;
        const int32_t l = bestLabel;
        PeelMode pm;
        pm.extent = bestExtent;
        const int ia = idx(inner.A, l), ib = idx(inner.B, l), ic = idx(inner.C, l), id = idx(inner.D, l);
        if (ia >= 0) { pm.sA = inner.A.desc.stride[(size_t)ia]; inner.A.desc.extent[(size_t)ia] = 1; }
        if (ib >= 0) { pm.sB = inner.B.desc.stride[(size_t)ib]; inner.B.desc.extent[(size_t)ib] = 1; }
        if (ic >= 0) { pm.sC = inner.C.desc.stride[(size_t)ic]; inner.C.desc.extent[(size_t)ic] = 1; }
        if (id >= 0) { pm.sD = inner.D.desc.stride[(size_t)id]; inner.D.desc.extent[(size_t)id] = 1; }
        pm.contracted = (ic < 0);
        peel.push_back(pm);
    }
    return false;
}
// Contractions with a mode that ONE input carries and nothing else ('ijk,kl->il': j).  cutensorCreateContraction's GEMM view has no
// group for such a mode, but the reference's N-ary front end builds exactly these steps — _compute_target_tensor drops every mode no
// later operand or the target needs (cuTENSOR/python/cutensor/torch/einsum.py:111-156) — and torch.einsum, its comparator, sums the
// mode away.  Here: the operand is reduced over its lone modes first (cutensorReduce, OP_ADD) into a packed temporary in the workspace,
// then the ordinary contraction runs on the temporary.  (Summing first is also the cheap order: the contraction shrinks by the mode's
// extent.)  Returns false when the descriptor has no such mode; fills `inner` (the contraction on the temporaries) and the
// reductions otherwise.
struct LoneSplit {
    cutensorOperationDescriptor inner, redA, redB;
    bool hasA = false, hasB = false;
    uint64_t bytesA = 0, bytesB = 0;       // packed sizes of the temporaries
};
static bool split_lone_modes(const cutensorOperationDescriptor& desc, LoneSplit& out) {
    auto has = [](const std::vector<int32_t>& v, int32_t l) { return std::find(v.begin(), v.end(), l) != v.end(); };
    auto reduce_operand = [&](const TensorUse& X, const TensorUse& other, cutensorOperationDescriptor& red, TensorUse& kept, uint64_t& bytes) {
        bool lone = false;
        for (size_t i = 0; i < X.modes.size(); ++i)
            if (X.desc.extent[i] != 1 && !has(other.modes, X.modes[i]) && !has(desc.C.modes, X.modes[i])) lone = true;
        if (!lone) return false;
        kept = TensorUse{};
        kept.present = true;
        kept.op = X.op;                                         // conj(sum) = sum(conj): the inner contraction conjugates
        kept.desc.dtype = X.desc.dtype;
        kept.desc.alignment = 128;                              // a piece of the workspace at a multiple of 256 bytes: as aligned as the workspace itself (contraction.cu:242 asserts 128)
        int64_t run = 1;
        for (size_t i = 0; i < X.modes.size(); ++i) {
            if (!has(other.modes, X.modes[i]) && !has(desc.C.modes, X.modes[i])) continue;   // summed away (extent-1 lone modes too)
            kept.modes.push_back(X.modes[i]);
            kept.desc.extent.push_back(X.desc.extent[i]);
            kept.desc.stride.push_back(run);
            run *= X.desc.extent[i];
        }
        kept.desc.numModes = (uint32_t)kept.modes.size();
        bytes = (uint64_t)run * dtype_size(X.desc.dtype);
        red = cutensorOperationDescriptor{};
        red.kind = OpKind::Reduction;
        red.A = X;
        red.A.op = CUTENSOR_OP_IDENTITY;
        red.C = kept; red.C.op = CUTENSOR_OP_IDENTITY;
        red.D = red.C;
        red.opReduce = CUTENSOR_OP_ADD;
        red.compute = desc.compute;
        red.scalarType = desc.scalarType;
        return true;
    };
    TensorUse keptA, keptB;
    out.hasA = reduce_operand(desc.A, desc.B, out.redA, keptA, out.bytesA);
    out.hasB = reduce_operand(desc.B, desc.A, out.redB, keptB, out.bytesB);
    if (!out.hasA && !out.hasB) return false;
    out.inner = desc;
    if (out.hasA) out.inner.A = keptA;
    if (out.hasB) out.inner.B = keptB;
    return true;
}

// Repacked operands (round 6).  A 16-bit contraction whose operands the LDS-DMA kernels cannot stage in 16-byte units — an operand that is
// contiguous in a mode the K order does not start with ('ijk,lkj->il': A contiguous in k, B in j; the reference's 'mlik,lkjm->lij'), or in
// no staged mode at all — runs on the general family at 2-byte gathers: 70 TFLOP/s at 4096^2 x 1152 where the vendor BLAS reaches 630.
// When the problem is large enough to pay for it, the operand is copied FIRST (cutensorPermute, ~5 TB/s) into a packed temporary in the
// workspace — contracted modes fastest, in the order of the other operand's strides, then its free modes in D's order, then the batch
// modes — and the contraction runs on the temporaries on the LDS-DMA kernels.  (The reference library's own heuristics are closed; its
// samples only require that any stride pattern is accepted: cuTENSOR/contraction.cu:33-59.)  Decided by a time model: the copies at
// 4 TB/s + 4 us each + the LDS-DMA plan's estimate against the general family at 100 TFLOP/s (an operand on 2-byte gathers) or 600.
// several contracted modes and the fastest one fills less than 70 % of its K-tiles (the sweep mask of the LDS-DMA kernels keeps such a
// problem, plan_contraction.cpp h16_sweep_ragged, at that efficiency)
static bool h16_sweep_waste(const ContractionView& v) {
    if (v.K.size() < 2) return false;
    const int64_t e0 = v.K.front().extent;
    return e0 % 64 != 0 && (double)e0 < 0.7 * 64.0 * (double)((e0 + 63) / 64);
}
// fp32: what the plan that takes the operands as they lie will cost, when it is NOT on the LDS-DMA ring kernels (0: it is, or nothing to
// compare) — the cost model's estimate of the register-staged kernels, corrected by what they measure on the shapes of
// profiles/r06zzb_sweep_shapes_f32.jsonl: x 1.15 with 16-byte lanes (422 / 2103 / 170 us modelled, 498 / 2333 / 204 measured), x 2.5 when an
// operand is gathered element by element (390 / 265 / 1346 modelled, 1064 / 552 / 5865 measured)
static double f32_direct_estimate_us(const ContractionView& v, const std::vector<ContractionChoice>& ch) {
    if (v.dtype != HIP_R_32F || v.wide || ch.empty() || ch[0].family != 0 || ch[0].kernel < 0) return 0.0;
    int cnt = 0;
    const GettKernelInfo* t32 = gett_f32_kernels(&cnt);
    if (ch[0].kernel >= cnt || t32[ch[0].kernel].fragPartials) return 0.0;
    return ch[0].estimateUs * ((v.layA == LAY_S || v.layB == LAY_S) ? 2.5 : 1.15);
}
// fp64: what the general family costs when the direct plan gathers single elements (V = 1: 'ijk,lkj->il' 30 TFLOP/s, the larger 'mlik' case
// 15 — profiles/r06zzi_sweep_shapes_f64.jsonl); 0 when it stages 16-byte units (nothing to gain from a copy)
// complex64 likewise (8 real flops per multiply-add): 103 / 58 TFLOP/s on those two shapes at V = 1 (profiles/r06zzm_sweep_shapes_c64_before.jsonl)
static double f64_direct_estimate_us(const ContractionView& v, const ContractionChoice& gc) {
    if ((v.dtype != HIP_R_64F && v.dtype != HIP_C_32F) || v.wide || gc.family != 2 || gc.kernel < 0) return 0.0;
    int cnt = 0;
    const GettKernelInfo* tg = gett_gen_kernels(&cnt);
    if (gc.kernel >= cnt || tg[gc.kernel].vec >= 2) return 0.0;
    const double mnk = (double)v.totL * (double)v.totM * (double)v.totN * (double)v.totK;
    return (v.dtype == HIP_C_32F ? 8.0 * mnk / 75e12 : 2.0 * mnk / 22e12) * 1e6 + 8.0;
}
// set while the inner contraction of a repacked plan is estimated / planned: the temporaries are final, no second round of copies
static thread_local bool t_inRepack = false;
struct RepackScope { bool prev; RepackScope() : prev(t_inRepack) { t_inRepack = true; } ~RepackScope() { t_inRepack = prev; } };
struct RepackSplit {
    cutensorOperationDescriptor inner, permA, permB;
    bool hasA = false, hasB = false;
    uint64_t bytesA = 0, bytesB = 0;       // packed sizes of the temporaries
};
// tDirectUs: the estimate of the plan that takes the operands as they lie, when the LDS-DMA family has one (sweeps of a short ragged contracted
// mode waste most of every K-tile: 'abcd,dcbe->ae' with d = 16 keeps 16 of 64 k) — negative: the general family's model above.
static bool plan_repack(const cutensorHandle* handle, const cutensorOperationDescriptor& desc, const ContractionView& v, uint64_t wsLimit, double tDirectUs, RepackSplit& out) {
    const bool f32 = v.dtype == HIP_R_32F, c32 = v.dtype == HIP_C_32F, f64 = v.dtype == HIP_R_64F || c32;   // (f64: the general family's wide elements)
    if (t_inRepack || v.wide || (v.dtype != HIP_R_16BF && v.dtype != HIP_R_16F && !f32 && !f64) || v.K.empty()) return false;
    const double es = (double)dtype_size(v.dtype);
    if (desc.A.op != CUTENSOR_OP_IDENTITY || desc.B.op != CUTENSOR_OP_IDENTITY) return false;
    auto has = [](const std::vector<int32_t>& m, int32_t l) { return std::find(m.begin(), m.end(), l) != m.end(); };
    auto stride_of = [](const TensorUse& T, int32_t l) -> int64_t {
        for (size_t i = 0; i < T.modes.size(); ++i) if (T.modes[i] == l) return T.desc.stride[i];
        return 0;
    };
    // freeMajor: the FREE modes fastest (D's order), then the contracted modes in X's own order (a plain matrix transpose when they are
    // adjacent in X): the temporary is free-contiguous, and a free-contiguous operand takes ANY fastest contracted extent under the sweep
    // mask — the way out for sweeps that end in partial 16-byte units ('abcd,dcbe->ae' with d = 50)
    auto pack = [&](const TensorUse& X, const TensorUse& other, bool freeMajor, cutensorOperationDescriptor& perm, TensorUse& kept, uint64_t& bytes, double& copyUs) {
        struct Km { int32_t label; int64_t extent, so; };
        std::vector<Km> k;
        for (size_t i = 0; i < X.modes.size(); ++i) {
            const int32_t l = X.modes[i];
            const bool inO = has(other.modes, l), inD = has(desc.D.modes, l);
            if (!inO && !inD) return false;                                 // (an extent-1 mode nothing else carries: leave the descriptor alone)
            if (inO && !inD) k.push_back(Km{l, X.desc.extent[i], std::llabs(freeMajor ? X.desc.stride[i] : stride_of(other, l))});
        }
        std::stable_sort(k.begin(), k.end(), [](const Km& a, const Km& b) { return a.so < b.so; });
        kept = TensorUse{};
        kept.present = true;
        kept.op = X.op;
        kept.desc.dtype = X.desc.dtype;
        kept.desc.alignment = 128;                                          // (the workspace's own alignment, contraction.cu:242; the pieces start at multiples of 256 bytes)
        int64_t run = 1;
        auto push = [&](int32_t l, int64_t e) { kept.modes.push_back(l); kept.desc.extent.push_back(e); kept.desc.stride.push_back(run); run *= e; };
        if (!freeMajor) for (const Km& m : k) push(m.label, m.extent);
        for (int pass = 0; pass < 2; ++pass) {                              // free modes in D's order, then the batch modes
            if (freeMajor && pass == 1) for (const Km& m : k) push(m.label, m.extent);
            for (int32_t l : desc.D.modes) {
                if (!has(X.modes, l) || (has(other.modes, l) ? 1 : 0) != pass) continue;
                for (size_t i = 0; i < X.modes.size(); ++i) if (X.modes[i] == l) push(l, X.desc.extent[i]);
            }
        }
        if (kept.modes.size() != X.modes.size()) return false;
        kept.desc.numModes = (uint32_t)kept.modes.size();
        bytes = (uint64_t)run * dtype_size(X.desc.dtype);
        perm = cutensorOperationDescriptor{};
        perm.kind = OpKind::Permutation;
        perm.A = X;
        perm.D = kept; perm.D.op = CUTENSOR_OP_IDENTITY;
        perm.compute = desc.compute;
        perm.scalarType = desc.scalarType;
        perm.movedBytes = 2.0 * (double)bytes;
        // what the copy costs, by the kernel the element-wise planner gives it: the tiled kernels move whole tiles at ~4 TB/s (a tile mode
        // much shorter than its tile pays for the padding), the element-gather kernel ~15 G elements per second (measured, round 6:
        // 13 MB of bf16 from [d = 50, c, b, a] to [b, c, d, a] in 380 us)
        EwPlan ep;
        if (plan_elementwise(perm, ep, nullptr) != CUTENSOR_STATUS_SUCCESS) return false;
        const double elems = (double)run;
        if (ep.variant == EW_TRANSPOSE) {
            // (rows of a tile past the end of a mode are skipped, not moved: the padding costs about a third of live data — 'jkl -> kjl'
            // with 16 x 72 of every 64 x 128 tile pair live, 9.4 MB, measured ~15 us)
            const double padded = (double)ep.p.tiles0 * ep.p.tile0 * (double)ep.p.tiles1 * ep.p.tile1 * (double)ep.p.rest.total;
            copyUs = 4.0 + 2.0 * es * (elems + 0.35 * (padded - elems)) / 4e6;
        }
        else if (ep.variant == EW_ROWCOPY || ep.variant == EW_BLOCK) copyUs = 4.0 + 2.0 * es * elems / 4e6;
        else if (ep.variant == EW_TRANSPOSE_ANY) copyUs = 4.0 + 2.0 * es * elems / 2e6;
        else copyUs = 4.0 + elems / 15e3;
        return true;
    };
    const bool slowA = (v.swapped ? v.layB : v.layA) == LAY_S, slowB = (v.swapped ? v.layA : v.layB) == LAY_S;   // the user's A is kernel-B when swapped
    const double flops = 2.0 * (double)v.totL * (double)v.totM * (double)v.totN * (double)v.totK;
    const double tGeneral = tDirectUs >= 0.0 ? tDirectUs : flops / ((slowA || slowB) ? 100e12 : 400e12) * 1e6 + 8.0;
    // candidates: A, B or both copied, each with its contracted or its free modes fastest — with both K-major, the temporaries share one
    // order of the contracted modes, which then fuse into a single one (nothing ragged but the end of K).  The fastest one by the model,
    // if it beats the direct plan by a fifth.
    // (CUTENSOR_AMD_REPACK=f, hooks flavour: whenever the temporaries fit — the fuzzers' and the small parity cases' way onto this path)
    const bool forced = CTAMD_HOOK_ENV("CUTENSOR_AMD_REPACK") && CTAMD_HOOK_ENV("CUTENSOR_AMD_REPACK")[0] == 'f';
    double best = 1e30;
    for (int attempt = 1; attempt < 9; ++attempt) {                         // per operand: 0 = as it lies, 1 = contracted modes fastest, 2 = free modes fastest
        const int howA = attempt % 3, howB = attempt / 3;
        const bool doA = howA != 0, doB = howB != 0;
        // (every combination is tried: an operand the kernels cannot stage as it lies may become stageable once the OTHER one is copied —
        // A[d = 50, c, b, a] is K-contiguous in the fused mode (d, c, b) as soon as B holds the contracted modes in that order)
        RepackSplit r;
        TensorUse keptA, keptB;
        // the order of the contracted modes follows the OTHER operand as it will be contracted: with both repacked, B follows A's temporary
        double usA = 0.0, usB = 0.0;
        if (doA && !pack(desc.A, desc.B, howA == 2, r.permA, keptA, r.bytesA, usA)) continue;
        if (doB && !pack(desc.B, doA ? keptA : desc.A, howB == 2, r.permB, keptB, r.bytesB, usB)) continue;
        r.hasA = doA; r.hasB = doB;
        r.inner = desc;
        if (doA) r.inner.A = keptA;
        if (doB) r.inner.B = keptB;
        const uint64_t temps = ((r.bytesA + 255) & ~255ull) + ((r.bytesB + 255) & ~255ull);
        if (temps > wsLimit) continue;
        ContractionView vi;
        ContractionChoice hc;
        if (build_contraction_view(r.inner, vi, nullptr) != CUTENSOR_STATUS_SUCCESS || vi.wide) continue;
        if (f64) {
            // fp64 (general MFMA family, gett_gen.inc): the temporaries must give both operands 16-byte units (V = 2) where the direct plan
            // gathers single elements; the family's rates on the shapes of profiles/r06zzi_sweep_shapes_f64.jsonl: 52-59 TFLOP/s at V = 2
            int cnt = 0;
            const GettKernelInfo* tg = gett_gen_kernels(&cnt);
            if (!pick_gen_choice(vi, wsLimit - temps, handle->numCUs, hc) || hc.kernel < 0 || hc.kernel >= cnt || tg[hc.kernel].vec < 2) continue;
            hc.estimateUs = (c32 ? 4.0 * flops / 124e12 : flops / 55e12) * 1e6 + 8.0;   // (complex64 at V = 2: 124 TFLOP/s of real flops, r06zzm)
        } else if (f32) {
            // fp32: the temporaries must put the problem on the LDS-DMA ring kernels (gett_f32_stream.hip: whole 32-deep K-tiles in the
            // fastest contracted mode, or one ragged contracted mode)
            const std::vector<ContractionChoice> ci = rank_contraction_choices(vi, wsLimit - temps, handle->numCUs, false);
            int cnt = 0;
            const GettKernelInfo* t32 = gett_f32_kernels(&cnt);
            if (ci.empty() || ci[0].family != 0 || ci[0].kernel < 0 || ci[0].kernel >= cnt || (!forced && !t32[ci[0].kernel].fragPartials)) continue;
            hc = ci[0];
        } else if (!pick_h16_choice(vi, wsLimit - temps, handle->numCUs, hc)) continue;
        const double tCopies = usA + usB;
        if (tCopies + hc.estimateUs < best) { best = tCopies + hc.estimateUs; out = r; }
    }
    if (best >= 1e30) return false;
    return forced || best < 0.8 * tGeneral;
}

// pointer alignment the offset operands of a peeled contraction still have
static void peel_fix_alignment(cutensorOperationDescriptor& inner, const std::vector<PeelMode>& peel) {
    const int64_t es = (int64_t)dtype_size(inner.A.desc.dtype);
    auto fix = [&](TensorUse& t, int which) {
        uint32_t a = t.desc.alignment;
        for (const PeelMode& pm : peel) {
            const int64_t s = which == 0 ? pm.sA : which == 1 ? pm.sB : which == 2 ? pm.sC : pm.sD;
            while (a > (uint32_t)es && ((s * es) % (int64_t)a) != 0) a >>= 1;
        }
        t.desc.alignment = std::max<uint32_t>(a, (uint32_t)es);
    };
    fix(inner.A, 0); fix(inner.B, 1); fix(inner.C, 2); fix(inner.D, 3);
}

extern "C" {

// ---- handle (contraction.cu:123-124) -----------------------------------------------------------
cutensorStatus_t cutensorCreate(cutensorHandle_t* handle) try {
    if (handle == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorHandle* h = new (std::nothrow) cutensorHandle();
    if (h == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        h->device = dev;
        h->haveDevice = true;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            h->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            h->clockKHz = prop.clockRate > 0 ? prop.clockRate : 2400000;
            if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
                CT_LOG("warning: device is %s, kernels are built for gfx950", prop.gcnArchName);
        }
    } else {
        (void)hipGetLastError();   // no device: descriptor / planning calls still work
    }
    *handle = h;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

cutensorStatus_t cutensorDestroy(cutensorHandle_t handle) try {
    if (handle != nullptr) {
        for (auto& m : handle->pending) { (void)hipEventDestroy(m.e0); (void)hipEventDestroy(m.e1); }
        for (auto& ev : handle->prof.events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    }
    if (handle != nullptr && handle->syncPool != nullptr) (void)hipFree(handle->syncPool);
    delete handle;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// einsum.cu:445
cutensorStatus_t cutensorHandleResizePlanCache(cutensorHandle_t handle, const uint32_t numEntries) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    std::lock_guard<std::mutex> g(handle->mtx);
    handle->planCacheCapacity = numEntries;
    while (handle->planCache.size() > numEntries) handle->planCache.erase(handle->planCache.begin());
    if (handle->planMemo.size() > numEntries) handle->planMemo.clear();
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// contraction_plan_cache.cu:324-337 — one line per cached problem: key \t kernel \t splitK
cutensorStatus_t cutensorHandleWritePlanCacheToFile(const cutensorHandle_t handle, const char filename[]) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (filename == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    resolve_pending_measurements(handle);
    std::lock_guard<std::mutex> g(handle->mtx);
    FILE* f = std::fopen(filename, "w");
    if (f == nullptr) return CUTENSOR_STATUS_IO_ERROR;
    std::fprintf(f, "cutensor-amd-plancache 1\n");
    for (const auto& kv : handle->planCache) {   // key, kernel, splitK, candidates tried by incremental autotuning, best time [us]
        auto t = handle->tuning.find(kv.first);
        const int tried = t != handle->tuning.end() ? t->second.next : 0;
        const double us = (t != handle->tuning.end() && t->second.bestMs < 1e29f) ? t->second.bestMs * 1e3 : 0.0;
        std::fprintf(f, "%s\t%d\t%u\t%d\t%.3f\n", kv.second.key.c_str(), kv.second.kernel, kv.second.splitK, tried, us);
    }
    std::fclose(f);
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// contraction_plan_cache.cu:132-148
cutensorStatus_t cutensorHandleReadPlanCacheFromFile(cutensorHandle_t handle, const char filename[],
                                                     uint32_t* numCachelinesRead) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (filename == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (numCachelinesRead) *numCachelinesRead = 0;
    FILE* f = std::fopen(filename, "r");
    if (f == nullptr) return CUTENSOR_STATUS_IO_ERROR;
    char header[64];
    int version = 0;
    if (std::fscanf(f, "%63s %d\n", header, &version) != 2 || std::strcmp(header, "cutensor-amd-plancache") != 0) {
        std::fclose(f);
        return CUTENSOR_STATUS_IO_ERROR;
    }
    std::lock_guard<std::mutex> g(handle->mtx);
    handle->planMemo.clear();   // the file may bring a different tuned choice for a memoised problem
    std::vector<char> line(1 << 16);
    uint32_t n = 0;
    while (std::fgets(line.data(), (int)line.size(), f)) {
        char* t1 = std::strchr(line.data(), '\t');
        if (!t1) continue;
        *t1 = 0;
        PlanCacheEntry e;
        e.key = line.data();
        unsigned sk = 1;
        int tried = 0;
        double us = 0.0;
        if (std::sscanf(t1 + 1, "%d\t%u\t%d\t%lf", &e.kernel, &sk, &tried, &us) < 2) continue;
        e.splitK = sk;
        if (handle->planCache.size() >= handle->planCacheCapacity && !handle->planCache.count(e.key)) {
            std::fclose(f);
            if (numCachelinesRead) *numCachelinesRead = n;
            return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE;   // cache too small for the file
        }
        handle->planCache[e.key] = e;
        if (tried > 0) {   // resume incremental autotuning where the writing process stopped
            cutensorHandle::TuneState& t = handle->tuning[e.key];
            t.next = tried; t.bestKernel = e.kernel; t.bestSplitK = e.splitK; t.bestMs = us > 0.0 ? (float)(us * 1e-3) : 1e30f;
        }
        ++n;
    }
    std::fclose(f);
    if (numCachelinesRead) *numCachelinesRead = n;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// ---- tensor descriptor (contraction.cu:131-137) ------------------------------------------------
cutensorStatus_t cutensorCreateTensorDescriptor(const cutensorHandle_t handle, cutensorTensorDescriptor_t* desc,
                                                const uint32_t numModes, const int64_t extent[],
                                                const int64_t stride[], cutensorDataType_t dataType,
                                                uint32_t alignmentRequirement) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || (numModes > 0 && extent == nullptr)) return CUTENSOR_STATUS_INVALID_VALUE;
    if (numModes > 64) return CUTENSOR_STATUS_NOT_SUPPORTED;
    if (!supported_dtype(dataType)) return CUTENSOR_STATUS_NOT_SUPPORTED;
    const size_t es = dtype_size(dataType);
    if (alignmentRequirement == 0 || alignmentRequirement % es != 0) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorTensorDescriptor* d = new (std::nothrow) cutensorTensorDescriptor();
    if (d == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    d->numModes = numModes;
    d->dtype = dataType;
    d->alignment = alignmentRequirement;
    d->extent.assign(extent, extent + numModes);
    d->stride.resize(numModes);
    int64_t run = 1;
    for (uint32_t i = 0; i < numModes; ++i) {
        if (extent[i] <= 0) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
        if (stride != nullptr) {
            if (stride[i] <= 0) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
            d->stride[i] = stride[i];
        } else {
            d->stride[i] = run;   // packed generalized column-major (blocksparse.cu:80-81)
            run *= extent[i];
        }
    }
    *desc = d;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

cutensorStatus_t cutensorDestroyTensorDescriptor(cutensorTensorDescriptor_t desc) try {
    delete desc;   // NULL tolerated (python/einsum.h:302,396)
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// ---- operation descriptors ----------------------------------------------------------------------
static cutensorStatus_t new_op(cutensorOperationDescriptor_t* out, cutensorOperationDescriptor& tmp) {
    cutensorOperationDescriptor* o = new (std::nothrow) cutensorOperationDescriptor(tmp);
    if (o == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    *out = o;
    return CUTENSOR_STATUS_SUCCESS;
}

// contraction.cu:162-168
cutensorStatus_t cutensorCreateContraction(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                           const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                           const cutensorTensorDescriptor_t descB, const int32_t modeB[], cutensorOperator_t opB,
                                           const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                           const cutensorTensorDescriptor_t descD, const int32_t modeD[],
                                           const cutensorComputeDescriptor_t descCompute) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || !valid_compute(descCompute)) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorOperationDescriptor op{};
    op.kind = OpKind::Contraction;
    cutensorStatus_t st;
    if ((st = fill_use(op.A, descA, modeA, opA)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.B, descB, modeB, opB)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.C, descC, modeC, opC)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.D, descD, modeD, CUTENSOR_OP_IDENTITY)) != CUTENSOR_STATUS_SUCCESS) return st;
    op.compute = descCompute;
    op.scalarType = scalar_type_for(op.A.desc.dtype, descCompute);
    ContractionView v;
    std::string why;
    {
        LoneSplit ls;     // a mode only one input carries: validated as the reduction(s) + the contraction on the temporaries (split_lone_modes)
        if (split_lone_modes(op, ls)) {
            ReducePlan rp;
            if (ls.hasA && (st = plan_reduction(ls.redA, 0, handle->numCUs, rp, &why)) != CUTENSOR_STATUS_SUCCESS) { CT_LOG("cutensorCreateContraction: %s", why.c_str()); return st; }
            if (ls.hasB && (st = plan_reduction(ls.redB, 0, handle->numCUs, rp, &why)) != CUTENSOR_STATUS_SUCCESS) { CT_LOG("cutensorCreateContraction: %s", why.c_str()); return st; }
            st = build_contraction_view(ls.inner, v, &why);
            if (st != CUTENSOR_STATUS_SUCCESS) { CT_LOG("cutensorCreateContraction: %s", why.c_str()); return st; }
            op.flops = 2.0 * (double)v.totL * (double)v.totM * (double)v.totN * (double)v.totK + (ls.hasA ? num_elements(op.A.desc) : 0.0) +
                       (ls.hasB ? num_elements(op.B.desc) : 0.0);
            const double esz = (double)dtype_size(op.A.desc.dtype);
            op.movedBytes = esz * (num_elements(op.A.desc) + num_elements(op.B.desc) + num_elements(op.D.desc));
            return new_op(desc, op);
        }
    }
    st = build_contraction_view(op, v, &why);
    if (st != CUTENSOR_STATUS_SUCCESS) { CT_LOG("cutensorCreateContraction: %s", why.c_str()); return st; }
    // contraction.cu:61 / :274-276
    op.flops = 2.0 * (double)v.totL * (double)v.totM * (double)v.totN * (double)v.totK;
    const double es = (double)dtype_size(op.A.desc.dtype);
    op.movedBytes = es * (num_elements(op.A.desc) + num_elements(op.B.desc) + num_elements(op.D.desc));
    return new_op(desc, op);
} CTAMD_API_CATCH

// contraction_trinary.cu:191-198: E = alpha * A * B * C + beta * D, executed as two pairwise contractions through a
// packed intermediate T.  The pair contracted first is the one that minimises flops(first) + flops(second)
// (the sample itself states its flop count as that sum, :65-67); T keeps every mode of the pair that the third
// operand or the output still needs, in the order they appear in X then Y.
cutensorStatus_t cutensorCreateContractionTrinary(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                                  const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                                  const cutensorTensorDescriptor_t descB, const int32_t modeB[], cutensorOperator_t opB,
                                                  const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                                  const cutensorTensorDescriptor_t descD, const int32_t modeD[], cutensorOperator_t opD,
                                                  const cutensorTensorDescriptor_t descE, const int32_t modeE[],
                                                  const cutensorComputeDescriptor_t descCompute) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || !valid_compute(descCompute)) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorOperationDescriptor op{};
    op.kind = OpKind::ContractionTrinary;
    cutensorStatus_t st;
    if ((st = fill_use(op.A, descA, modeA, opA)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.B, descB, modeB, opB)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.C, descC, modeC, opC)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.D, descD, modeD, opD)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.E, descE, modeE, CUTENSOR_OP_IDENTITY)) != CUTENSOR_STATUS_SUCCESS) return st;
    op.compute = descCompute;
    op.scalarType = scalar_type_for(op.A.desc.dtype, descCompute);
    const TensorUse* in[3] = {&op.A, &op.B, &op.C};
    auto has = [](const std::vector<int32_t>& v, int32_t l) { return std::find(v.begin(), v.end(), l) != v.end(); };
    double bestCost = 0.0;
    bool found = false;
    static const int pairs[3][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}};
    for (const auto& pr : pairs) {
        const TensorUse &X = *in[pr[0]], &Y = *in[pr[1]], &Z = *in[pr[2]];
        // T: modes of X and Y still needed by Z or E
        TensorUse T;
        T.present = true;
        T.desc.dtype = X.desc.dtype;
        T.desc.alignment = 128;
        double flops1 = 2.0, tElems = 1.0;
        std::vector<int32_t> seen;
        auto visit = [&](const TensorUse& U) {
            for (size_t i = 0; i < U.modes.size(); ++i) {
                const int32_t l = U.modes[i];
                if (has(seen, l)) continue;
                seen.push_back(l);
                flops1 *= (double)U.desc.extent[i];
                if (has(Z.modes, l) || has(op.E.modes, l)) {
                    T.modes.push_back(l);
                    T.desc.extent.push_back(U.desc.extent[i]);
                    tElems *= (double)U.desc.extent[i];
                }
            }
        };
        visit(X); visit(Y);
        T.desc.numModes = (uint32_t)T.modes.size();
        T.desc.stride.resize(T.modes.size());
        int64_t run = 1;
        for (size_t i = 0; i < T.modes.size(); ++i) { T.desc.stride[i] = run; run *= T.desc.extent[i]; }
        double flops2 = 2.0;
        std::vector<int32_t> seen2;
        for (const TensorUse* U : {const_cast<const TensorUse*>(&T), &Z})
            for (size_t i = 0; i < U->modes.size(); ++i)
                if (!has(seen2, U->modes[i])) { seen2.push_back(U->modes[i]); flops2 *= (double)U->desc.extent[i]; }
        cutensorOperationDescriptor s1{}, s2{};
        s1.kind = s2.kind = OpKind::Contraction;
        s1.compute = s2.compute = descCompute;
        s1.scalarType = s2.scalarType = op.scalarType;
        s1.A = X; s1.B = Y; s1.C = T; s1.D = T;
        s2.A = T; s2.B = Z; s2.C = op.D; s2.D = op.E;
        ContractionView v1, v2;
        std::string why;
        if (build_contraction_view(s1, v1, &why) != CUTENSOR_STATUS_SUCCESS || build_contraction_view(s2, v2, &why) != CUTENSOR_STATUS_SUCCESS) continue;
        s1.flops = flops1; s2.flops = flops2;
        const double cost = flops1 + flops2 + 8.0 * tElems;   // the intermediate is written and read once
        if (!found || cost < bestCost) {
            found = true;
            bestCost = cost;
            op.sub.clear();
            op.sub.push_back(s1);
            op.sub.push_back(s2);
            op.triOrder[0] = pr[0]; op.triOrder[1] = pr[1]; op.triOrder[2] = pr[2];
            op.tBytes = (uint64_t)tElems * dtype_size(T.desc.dtype);
            op.flops = flops1 + flops2;
        }
    }
    if (!found) { CT_LOG("cutensorCreateContractionTrinary: no pairwise order is supported"); return CUTENSOR_STATUS_NOT_SUPPORTED; }
    const double es = (double)dtype_size(op.A.desc.dtype);
    op.movedBytes = es * (num_elements(op.A.desc) + num_elements(op.B.desc) + num_elements(op.C.desc) + num_elements(op.E.desc));
    return new_op(desc, op);
} CTAMD_API_CATCH

// reduction.cu:141-146
cutensorStatus_t cutensorCreateReduction(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                         const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                         const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                         const cutensorTensorDescriptor_t descD, const int32_t modeD[],
                                         cutensorOperator_t opReduce, const cutensorComputeDescriptor_t descCompute) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || !valid_compute(descCompute)) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorOperationDescriptor op{};
    op.kind = OpKind::Reduction;
    cutensorStatus_t st;
    if ((st = fill_use(op.A, descA, modeA, opA)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.C, descC, modeC, opC)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.D, descD, modeD, CUTENSOR_OP_IDENTITY)) != CUTENSOR_STATUS_SUCCESS) return st;
    op.opReduce = opReduce;
    op.compute = descCompute;
    op.scalarType = scalar_type_for(op.A.desc.dtype, descCompute);
    ReducePlan rp;
    std::string why;
    st = plan_reduction(op, 0, handle->numCUs, rp, &why);   // validates the problem
    if (st != CUTENSOR_STATUS_SUCCESS) { CT_LOG("cutensorCreateReduction: %s", why.c_str()); return st; }
    const double es = (double)dtype_size(op.A.desc.dtype);
    op.flops = num_elements(op.A.desc);
    op.movedBytes = es * (num_elements(op.A.desc) + num_elements(op.D.desc));   // reduction.cu:229-231
    return new_op(desc, op);
} CTAMD_API_CATCH

// elementwise_permute.cu:142-149
cutensorStatus_t cutensorCreatePermutation(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                           const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                           const cutensorTensorDescriptor_t descB, const int32_t modeB[],
                                           const cutensorComputeDescriptor_t descCompute) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || !valid_compute(descCompute)) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorOperationDescriptor op{};
    op.kind = OpKind::Permutation;
    cutensorStatus_t st;
    if ((st = fill_use(op.A, descA, modeA, opA)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.D, descB, modeB, CUTENSOR_OP_IDENTITY)) != CUTENSOR_STATUS_SUCCESS) return st;
    op.compute = descCompute;
    op.scalarType = scalar_type_for(op.A.desc.dtype, descCompute);
    EwPlan ep;
    std::string why;
    st = plan_elementwise(op, ep, &why);
    if (st != CUTENSOR_STATUS_SUCCESS) { CT_LOG("cutensorCreatePermutation: %s", why.c_str()); return st; }
    op.movedBytes = 2.0 * (double)dtype_size(op.D.desc.dtype) * num_elements(op.D.desc);   // elementwise_permute.cu:208
    return new_op(desc, op);
} CTAMD_API_CATCH

// elementwise_binary.cu:149-153 — only opAC = ADD is implemented
cutensorStatus_t cutensorCreateElementwiseBinary(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                                 const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                                 const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                                 const cutensorTensorDescriptor_t descD, const int32_t modeD[],
                                                 cutensorOperator_t opAC, const cutensorComputeDescriptor_t descCompute) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || !valid_compute(descCompute)) return CUTENSOR_STATUS_INVALID_VALUE;
    if (opAC != CUTENSOR_OP_ADD && opAC != CUTENSOR_OP_MUL && opAC != CUTENSOR_OP_MAX && opAC != CUTENSOR_OP_MIN) return CUTENSOR_STATUS_NOT_SUPPORTED;
    cutensorOperationDescriptor op{};
    op.kind = OpKind::ElementwiseBinary;
    cutensorStatus_t st;
    if ((st = fill_use(op.A, descA, modeA, opA)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.C, descC, modeC, opC)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.D, descD, modeD, CUTENSOR_OP_IDENTITY)) != CUTENSOR_STATUS_SUCCESS) return st;
    op.opReduce = opAC;
    op.compute = descCompute;
    op.scalarType = scalar_type_for(op.A.desc.dtype, descCompute);
    EwPlan ep;
    std::string why;
    st = plan_elementwise(op, ep, &why);
    if (st != CUTENSOR_STATUS_SUCCESS) { CT_LOG("cutensorCreateElementwiseBinary: %s", why.c_str()); return st; }
    op.movedBytes = 3.0 * (double)dtype_size(op.D.desc.dtype) * num_elements(op.D.desc);
    return new_op(desc, op);
} CTAMD_API_CATCH

// elementwise_trinary.cu:174-182
cutensorStatus_t cutensorCreateElementwiseTrinary(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                                  const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                                  const cutensorTensorDescriptor_t descB, const int32_t modeB[], cutensorOperator_t opB,
                                                  const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                                  const cutensorTensorDescriptor_t descD, const int32_t modeD[],
                                                  cutensorOperator_t opAB, cutensorOperator_t opABC,
                                                  const cutensorComputeDescriptor_t descCompute) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || !valid_compute(descCompute)) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorOperationDescriptor op{};
    op.kind = OpKind::ElementwiseTrinary;
    cutensorStatus_t st;
    if ((st = fill_use(op.A, descA, modeA, opA)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.B, descB, modeB, opB)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.C, descC, modeC, opC)) != CUTENSOR_STATUS_SUCCESS) return st;
    if ((st = fill_use(op.D, descD, modeD, CUTENSOR_OP_IDENTITY)) != CUTENSOR_STATUS_SUCCESS) return st;
    op.opAB = opAB;
    op.opReduce = opABC;
    op.compute = descCompute;
    op.scalarType = scalar_type_for(op.A.desc.dtype, descCompute);
    EwTrinaryPlan tp;
    std::string why;
    st = plan_elementwise_trinary(op, tp, &why);
    if (st != CUTENSOR_STATUS_SUCCESS) { CT_LOG("cutensorCreateElementwiseTrinary: %s", why.c_str()); return st; }
    op.movedBytes = 4.0 * (double)dtype_size(op.D.desc.dtype) * num_elements(op.D.desc);   // elementwise_trinary.cu:234-238
    return new_op(desc, op);
} CTAMD_API_CATCH

cutensorStatus_t cutensorDestroyOperationDescriptor(cutensorOperationDescriptor_t desc) try {
    delete desc;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// contraction.cu:176-180, contraction_jit.cu:379-383
cutensorStatus_t cutensorOperationDescriptorGetAttribute(const cutensorHandle_t handle, cutensorOperationDescriptor_t desc,
                                                         cutensorOperationDescriptorAttribute_t attr, void* buf,
                                                         size_t sizeInBytes) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || buf == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    switch (attr) {
        case CUTENSOR_OPERATION_DESCRIPTOR_TAG:
            if (sizeInBytes != sizeof(int32_t)) return CUTENSOR_STATUS_INVALID_VALUE;
            *static_cast<int32_t*>(buf) = desc->tag;
            return CUTENSOR_STATUS_SUCCESS;
        case CUTENSOR_OPERATION_DESCRIPTOR_SCALAR_TYPE:
            if (sizeInBytes != sizeof(cutensorDataType_t)) return CUTENSOR_STATUS_INVALID_VALUE;
            *static_cast<cutensorDataType_t*>(buf) = desc->scalarType;
            return CUTENSOR_STATUS_SUCCESS;
        case CUTENSOR_OPERATION_DESCRIPTOR_FLOPS:
            if (sizeInBytes != sizeof(float)) return CUTENSOR_STATUS_INVALID_VALUE;
            *static_cast<float*>(buf) = (float)desc->flops;
            return CUTENSOR_STATUS_SUCCESS;
        case CUTENSOR_OPERATION_DESCRIPTOR_MOVED_BYTES:
            if (sizeInBytes != sizeof(float)) return CUTENSOR_STATUS_INVALID_VALUE;
            *static_cast<float*>(buf) = (float)desc->movedBytes;
            return CUTENSOR_STATUS_SUCCESS;
        default:
            return CUTENSOR_STATUS_NOT_SUPPORTED;
    }
} CTAMD_API_CATCH

cutensorStatus_t cutensorOperationDescriptorSetAttribute(const cutensorHandle_t handle, cutensorOperationDescriptor_t desc,
                                                         cutensorOperationDescriptorAttribute_t attr, const void* buf,
                                                         size_t sizeInBytes) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || buf == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (attr == CUTENSOR_OPERATION_DESCRIPTOR_TAG && sizeInBytes == sizeof(int32_t)) {
        desc->tag = *static_cast<const int32_t*>(buf);
        return CUTENSOR_STATUS_SUCCESS;
    }
    // elementwise_permute_padding.cu:178-195: one int per output mode / one output-typed value
    if (attr == CUTENSOR_OPERATION_DESCRIPTOR_PADDING_LEFT || attr == CUTENSOR_OPERATION_DESCRIPTOR_PADDING_RIGHT) {
        if (desc->kind != OpKind::Permutation) return CUTENSOR_STATUS_NOT_SUPPORTED;
        if (sizeInBytes != sizeof(int32_t) * desc->D.modes.size()) return CUTENSOR_STATUS_INVALID_VALUE;
        const int32_t* v = static_cast<const int32_t*>(buf);
        std::vector<int32_t>& dst = (attr == CUTENSOR_OPERATION_DESCRIPTOR_PADDING_LEFT) ? desc->padLeft : desc->padRight;
        dst.assign(v, v + desc->D.modes.size());
        for (int32_t x : dst) if (x < 0) { dst.clear(); return CUTENSOR_STATUS_INVALID_VALUE; }
        return CUTENSOR_STATUS_SUCCESS;
    }
    if (attr == CUTENSOR_OPERATION_DESCRIPTOR_PADDING_VALUE) {
        if (desc->kind != OpKind::Permutation) return CUTENSOR_STATUS_NOT_SUPPORTED;
        if (sizeInBytes != dtype_size(desc->D.desc.dtype)) return CUTENSOR_STATUS_INVALID_VALUE;
        switch (desc->D.desc.dtype) {
            case HIP_R_32F: desc->padValue = *static_cast<const float*>(buf); break;
            case HIP_R_64F: desc->padValue = *static_cast<const double*>(buf); break;
            case HIP_R_16F: { uint16_t u; std::memcpy(&u, buf, 2); const uint32_t s = (u >> 15) & 1u, e = (u >> 10) & 31u, m = u & 1023u;
                              double x = (e == 0) ? std::ldexp((double)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : std::ldexp((double)(m | 1024u), (int)e - 25));
                              desc->padValue = s ? -x : x; break; }
            case HIP_R_16BF: { uint16_t u; std::memcpy(&u, buf, 2); const uint32_t w = (uint32_t)u << 16; float f; std::memcpy(&f, &w, 4); desc->padValue = f; break; }
            default: return CUTENSOR_STATUS_NOT_SUPPORTED;
        }
        return CUTENSOR_STATUS_SUCCESS;
    }
    return CUTENSOR_STATUS_NOT_SUPPORTED;
} CTAMD_API_CATCH

// ---- plan preference (contraction.cu:194-198) ----------------------------------------------------
cutensorStatus_t cutensorCreatePlanPreference(const cutensorHandle_t handle, cutensorPlanPreference_t* pref,
                                              cutensorAlgo_t algo, cutensorJitMode_t jitMode) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (pref == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorPlanPreference* p = new (std::nothrow) cutensorPlanPreference();
    if (p == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    p->algo = algo;
    p->jit = jitMode;   // accepted and ignored: every kernel is ahead-of-time compiled for gfx950
    *pref = p;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

cutensorStatus_t cutensorDestroyPlanPreference(cutensorPlanPreference_t pref) try {
    delete pref;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// contraction_plan_cache.cu:215-237
cutensorStatus_t cutensorPlanPreferenceSetAttribute(const cutensorHandle_t handle, cutensorPlanPreference_t pref,
                                                    cutensorPlanPreferenceAttribute_t attr, const void* buf,
                                                    size_t sizeInBytes) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (pref == nullptr || buf == nullptr || sizeInBytes != 4) return CUTENSOR_STATUS_INVALID_VALUE;
    const int32_t v = *static_cast<const int32_t*>(buf);
    switch (attr) {
        case CUTENSOR_PLAN_PREFERENCE_AUTOTUNE_MODE: pref->autotune = (cutensorAutotuneMode_t)v; break;
        case CUTENSOR_PLAN_PREFERENCE_CACHE_MODE: pref->cacheMode = (cutensorCacheMode_t)v; break;
        case CUTENSOR_PLAN_PREFERENCE_INCREMENTAL_COUNT: pref->incrementalCount = v; break;
        case CUTENSOR_PLAN_PREFERENCE_ALGO: pref->algo = (cutensorAlgo_t)v; break;
        case CUTENSOR_PLAN_PREFERENCE_KERNEL_RANK: pref->kernelRank = v; break;
        case CUTENSOR_PLAN_PREFERENCE_JIT: pref->jit = (cutensorJitMode_t)v; break;
        case CUTENSOR_AMD_PLAN_PREFERENCE_OPERANDS_STREAMED: pref->operandsStreamed = v != 0 ? 1 : 0; break;   // engine extension
        default: return CUTENSOR_STATUS_INVALID_VALUE;
    }
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// contraction.cu:207-211
cutensorStatus_t cutensorEstimateWorkspaceSize(const cutensorHandle_t handle, const cutensorOperationDescriptor_t desc,
                                               const cutensorPlanPreference_t planPref,
                                               const cutensorWorksizePreference_t workspacePref,
                                               uint64_t* workspaceSizeEstimate) try {
    (void)planPref;
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || workspaceSizeEstimate == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    *workspaceSizeEstimate = 0;
    if (desc->kind == OpKind::BlockSparseContraction) return blocksparse_estimate(handle, *desc, workspaceSizeEstimate);
    if (desc->kind == OpKind::ContractionTrinary) {   // intermediate + the larger of the two pairwise needs
        uint64_t w1 = 0, w2 = 0;
        cutensorOperationDescriptor s1 = desc->sub[0], s2 = desc->sub[1];
        cutensorStatus_t st = cutensorEstimateWorkspaceSize(handle, &s1, planPref, workspacePref, &w1);
        if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorEstimateWorkspaceSize(handle, &s2, planPref, workspacePref, &w2);
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
        *workspaceSizeEstimate = ((desc->tBytes + 255) & ~255ull) + std::max(w1, w2);
        return CUTENSOR_STATUS_SUCCESS;
    }
    if (workspacePref == CUTENSOR_WORKSPACE_MIN) return CUTENSOR_STATUS_SUCCESS;
    const uint64_t cap = (workspacePref == CUTENSOR_WORKSPACE_MAX) ? (4ull << 30) : (1ull << 30);
    if (desc->kind == OpKind::Contraction) {
        {
            LoneSplit ls;     // the temporaries + the largest need of the reductions and the inner contraction
            if (split_lone_modes(*desc, ls)) {
                uint64_t wI = 0, wA = 0, wB = 0;
                cutensorStatus_t st = cutensorEstimateWorkspaceSize(handle, &ls.inner, planPref, workspacePref, &wI);
                if (st == CUTENSOR_STATUS_SUCCESS && ls.hasA) st = cutensorEstimateWorkspaceSize(handle, &ls.redA, planPref, workspacePref, &wA);
                if (st == CUTENSOR_STATUS_SUCCESS && ls.hasB) st = cutensorEstimateWorkspaceSize(handle, &ls.redB, planPref, workspacePref, &wB);
                if (st != CUTENSOR_STATUS_SUCCESS) return st;
                *workspaceSizeEstimate = ((ls.bytesA + 255) & ~255ull) + ((ls.bytesB + 255) & ~255ull) + std::max(wI, std::max(wA, wB));
                return CUTENSOR_STATUS_SUCCESS;
            }
        }
        ContractionView v;
        cutensorStatus_t st = build_contraction_view(*desc, v, nullptr);
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
        if (!v.wide && (v.dtype == HIP_R_16BF || v.dtype == HIP_R_16F)) {   // split-K partials of the 16-bit MFMA kernel
            ContractionChoice hc;
            RepackSplit rs;
            const bool direct = pick_h16_choice(v, cap, handle->numCUs, hc);
            const bool h16ok = desc->scalarType == HIP_R_32F && !(desc->compute && desc->compute->id == 5);
            if (direct && !(h16ok && h16_sweep_waste(v))) *workspaceSizeEstimate = hc.workspace;
            else if (h16ok && plan_repack(handle, *desc, v, cap, direct ? hc.estimateUs : -1.0, rs)) {
                // operands copied into packed temporaries first (plan_repack): the temporaries + what the copies and the inner contraction want
                const uint64_t temps = ((rs.bytesA + 255) & ~255ull) + ((rs.bytesB + 255) & ~255ull);
                uint64_t wI = 0;
                RepackScope scope;
                cutensorStatus_t st2 = cutensorEstimateWorkspaceSize(handle, &rs.inner, planPref, workspacePref, &wI);
                if (st2 != CUTENSOR_STATUS_SUCCESS) return st2;
                *workspaceSizeEstimate = temps + wI;
            } else if (direct || pick_gen_choice(v, cap, handle->numCUs, hc)) *workspaceSizeEstimate = hc.workspace;
            return CUTENSOR_STATUS_SUCCESS;
        }
        if (!v.wide && (v.dtype == HIP_R_64F || v.dtype == HIP_C_32F || v.dtype == HIP_C_64F)) {   // split-K partials of the general MFMA family
            ContractionChoice gc;
            if (pick_gen_choice(v, cap, handle->numCUs, gc)) {
                *workspaceSizeEstimate = gc.workspace;
                RepackSplit rs;     // fp64 on element gathers: an operand copied first when that pays (plan_repack)
                const double tDirect = desc->scalarType == v.dtype ? f64_direct_estimate_us(v, gc) : 0.0;
                if (tDirect > 0.0 && plan_repack(handle, *desc, v, cap, tDirect, rs)) {
                    const uint64_t temps = ((rs.bytesA + 255) & ~255ull) + ((rs.bytesB + 255) & ~255ull);
                    uint64_t wI = 0;
                    RepackScope scope;
                    cutensorStatus_t st2 = cutensorEstimateWorkspaceSize(handle, &rs.inner, planPref, workspacePref, &wI);
                    if (st2 != CUTENSOR_STATUS_SUCCESS) return st2;
                    *workspaceSizeEstimate = temps + wI;
                }
            }
            return CUTENSOR_STATUS_SUCCESS;
        }
        if (v.wide) {    // a peeled contraction wants what its inner, tiled problem wants
            cutensorOperationDescriptor inner;
            std::vector<PeelMode> peel;
            if (peel_wide_contraction(*desc, inner, peel)) {
                peel_fix_alignment(inner, peel);
                return cutensorEstimateWorkspaceSize(handle, &inner, planPref, workspacePref, workspaceSizeEstimate);
            }
            return CUTENSOR_STATUS_SUCCESS;
        }
        if (v.dtype != HIP_R_32F) return CUTENSOR_STATUS_SUCCESS;
        // the largest workspace any of the best few candidates would like to have
        std::vector<ContractionChoice> ch = rank_contraction_choices(v, cap, handle->numCUs, planPref != nullptr && planPref->operandsStreamed != 0);
        {
            RepackSplit rs;     // fp32 off the ring kernels: operands copied into packed temporaries first when that pays (plan_repack)
            const double tDirect = f32_direct_estimate_us(v, ch);
            if (tDirect > 0.0 && !(desc->compute && desc->compute->id == 5) && desc->scalarType == HIP_R_32F && plan_repack(handle, *desc, v, cap, tDirect, rs)) {
                const uint64_t temps = ((rs.bytesA + 255) & ~255ull) + ((rs.bytesB + 255) & ~255ull);
                uint64_t wI = 0;
                RepackScope scope;
                cutensorStatus_t st2 = cutensorEstimateWorkspaceSize(handle, &rs.inner, planPref, workspacePref, &wI);
                if (st2 != CUTENSOR_STATUS_SUCCESS) return st2;
                *workspaceSizeEstimate = temps + wI;
                return CUTENSOR_STATUS_SUCCESS;
            }
        }
        uint64_t want = 0;
        for (size_t i = 0; i < ch.size() && i < 4; ++i) want = std::max(want, ch[i].workspace);
        *workspaceSizeEstimate = want;
    } else if (desc->kind == OpKind::Reduction) {
        ReducePlan rp;
        cutensorStatus_t st = plan_reduction(*desc, cap, handle->numCUs, rp, nullptr);
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
        *workspaceSizeEstimate = rp.workspace;
    }
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// ---- measured selection for CUTENSOR_ALGO_DEFAULT_PATIENT -------------------------------------
// Times the best-ranked candidates on scratch tensors of the problem's own shape; plan creation is
// outside every timed region of the samples (contraction.cu:218-222 vs :253-270).
static int autotune_contraction(cutensorHandle_t handle, const cutensorOperationDescriptor& op,
                                const ContractionView& v, const std::vector<ContractionChoice>& ch) {
    if (ch.size() < 2) return 0;                 // nothing to choose between (the general MFMA family ranks ONE candidate)
    const int family = ch[0].family;             // 0 fp32 GETT, 1 aligned 16-bit (LDS-DMA), 2 general MFMA family — one table each
    for (const ContractionChoice& c : ch)
        if (c.family != family) return 0;        // mixed lists are not timed: a kernel index means nothing outside its own table
    const size_t es = dtype_size(op.A.desc.dtype);
    auto span = [&](const cutensorTensorDescriptor& d) {
        int64_t n = 1;
        for (uint32_t i = 0; i < d.numModes; ++i) n += (d.extent[i] - 1) * d.stride[i];
        return (size_t)n * es;
    };
    uint64_t wsMax = 0;
    const size_t nTry = std::min<size_t>(ch.size(), 12);
    for (size_t i = 0; i < nTry; ++i) wsMax = std::max(wsMax, ch[i].workspace);
    // scratch tensors and the event pair, released on every way out — an exception included (since round 5 the ABI turns bad_alloc into a
    // status code: what it unwinds through must not leak device memory in a process that keeps running)
    struct Scratch {
        void *A = nullptr, *B = nullptr, *D = nullptr, *W = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Scratch() {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            for (void* p : {A, B, D, W}) if (p) (void)hipFree(p);
        }
    } sc;
    void *&A = sc.A, *&B = sc.B, *&D = sc.D, *&W = sc.W;
    hipEvent_t &e0 = sc.e0, &e1 = sc.e1;
    int best = 0;
    if (hipMalloc(&A, span(op.A.desc)) != hipSuccess || hipMalloc(&B, span(op.B.desc)) != hipSuccess ||
        hipMalloc(&D, span(op.D.desc)) != hipSuccess || (wsMax && hipMalloc(&W, wsMax) != hipSuccess) ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        (void)hipGetLastError();
        return best;
    }
    (void)hipMemset(A, 0x3c, span(op.A.desc));   // 0x3c3c3c3c = 0.0115f (0x3c3c: 0.0115 in bf16, 1.06 in fp16): finite, non-trivial data
    (void)hipMemset(B, 0x3c, span(op.B.desc));
    {
        int count = 0;
        const GettKernelInfo* tab = family == 2 ? gett_gen_kernels(&count) : family == 1 ? gett_h16_kernels(&count) : gett_f32_kernels(&count);
        float bestMs = 1e30f;
        for (size_t i = 0; i < nTry; ++i) {
            GettParams gp;
            SplitKReduceParams rp;
            fill_gett_params(v, ch[i], gp, rp);
            gp.A = v.swapped ? B : A;
            gp.B = v.swapped ? A : B;
            gp.endA += (unsigned long long)(uintptr_t)gp.A;
            gp.endB += (unsigned long long)(uintptr_t)gp.B;
            gp.C = D; gp.D = D; gp.alpha = 1.f; gp.beta = 0.f;
            gp.partial = ch[i].splitK > 1 ? static_cast<float*>(W) : nullptr;
            rp.partial = static_cast<float*>(W); rp.C = D; rp.D = D; rp.alpha = 1.f; rp.beta = 0.f;
            // one launch = GETT kernel + (for split-K) the fold that matches the kernel's partial layout
            auto once = [&]() -> bool {
                if (tab[ch[i].kernel].launch(gp, nullptr) != hipSuccess) return false;
                if (ch[i].splitK > 1) {
                    // the fold that matches the kernel's partials: accumulator-register order (fp32 stream kernels), row-major fp32, or —
                    // general family with 8- / 16-byte elements — row-major partials in the accumulator type
                    const int elem = family == 2 ? tab[ch[i].kernel].elem : GEN_BF16;
                    const hipError_t e = (family == 2 && elem >= GEN_F64) ? launch_gen_splitk_reduce(rp, elem, nullptr)
                                       : tab[ch[i].kernel].fragPartials ? launch_splitk_reduce_frag(rp, nullptr) : launch_splitk_reduce(rp, nullptr);
                    if (e != hipSuccess) return false;
                }
                return true;
            };
            if (ch[i].kernel < 0 || ch[i].kernel >= count) continue;
            float ms = 1e30f;
            bool ok = once() && hipDeviceSynchronize() == hipSuccess;
            if (ok && i == 0) {
                // clock ramp: a cold MI355X runs the first ~15-20 ms of a kernel stream ~10 % slower than its steady state
                // (DESIGN.md section 6); without this the first candidates would be timed against a handicap
                (void)hipEventRecord(e0, nullptr);
                float spent = 0.f;
                for (int guard = 0; ok && spent < 20.f && guard < 4000; ++guard) {
                    for (int r = 0; r < 8 && ok; ++r) ok = once();
                    (void)hipEventRecord(e1, nullptr);
                    ok = ok && hipEventSynchronize(e1) == hipSuccess;
                    (void)hipEventElapsedTime(&spent, e0, e1);
                }
            }
            for (int rep = 0; ok && rep < 3; ++rep) {        // best of three batches of launches issued back to back
                const int batch = 8;
                (void)hipEventRecord(e0, nullptr);
                for (int r = 0; r < batch && ok; ++r) ok = once();
                (void)hipEventRecord(e1, nullptr);
                if (!ok || hipEventSynchronize(e1) != hipSuccess) { ok = false; break; }
                float t = 0.f;
                (void)hipEventElapsedTime(&t, e0, e1);
                ms = std::min(ms, t / batch);
            }
            if (!ok) { (void)hipGetLastError(); ms = 1e30f; }
            CT_LOG("autotune: cand %zu kernel %d (%dx%dx%d) splitK %u -> %.3f us (model %.1f us)", i, ch[i].kernel,
                   tab[ch[i].kernel].bm, tab[ch[i].kernel].bn, tab[ch[i].kernel].bk, ch[i].splitK, ms * 1e3, ch[i].estimateUs);
            if (ms < bestMs) { bestMs = ms; best = (int)i; }
        }
    }
    (void)handle;
    return best;
}

// ---- per-XCD K split for one-tile split-K plans ---------------------------------------------------------------
// The eight XCDs of an MI355X sustain slightly different shader clocks under the fp32-MFMA streaming kernel
// (2.00-2.07 GHz on the parts measured, stable per part, `profiles/r01b_phase_timing_einsum.json`), and with an
// even K split the slower dies finish 1-1.5 us after the faster ones.  Once per handle the plan's own kernel is
// run on scratch tensors with its in-kernel timestamps on; the per-XCD clock (shader cycles / wall time) becomes
// the weight of that XCD's share of K.  Returns the packed tiles-per-slice bytes, 0 = keep the uniform split.
static unsigned long long calibrate_xcd_split(cutensorHandle_t handle, const cutensorOperationDescriptor& op, const cutensorPlan& pl) {
    const uint32_t splitK = pl.gett.splitK;
    if (pl.gett.nBlocks != splitK || splitK % 8 != 0 || splitK > (uint32_t)handle->numCUs) return 0;
    int count = 0;
    const GettKernelInfo* tab = gett_f32_kernels(&count);
    const GettKernelInfo& k = tab[pl.choice.kernel];
    const uint64_t kTiles = pl.view.totK / k.bk;
    const uint64_t perXcdSum = kTiles / (splitK / 8);            // sum over x of tiles-per-slice(x)
    if (kTiles % (splitK / 8) != 0 || perXcdSum / 8 < 8 || perXcdSum / 8 > 200) return 0;
    {
        std::lock_guard<std::mutex> g(handle->mtx);
        if (handle->xcdSpeed.empty()) {
            handle->xcdSpeed.assign(8, 1.0);
            const size_t es = 4;
            auto span = [&](const cutensorTensorDescriptor& d) {
                int64_t n = 1;
                for (uint32_t i = 0; i < d.numModes; ++i) n += (d.extent[i] - 1) * d.stride[i];
                return (size_t)n * es;
            };
            void *A = nullptr, *B = nullptr, *D = nullptr, *W = nullptr;
            unsigned long long* T = nullptr;
            const size_t tBytes = (size_t)splitK * 16 * sizeof(unsigned long long);
            std::vector<unsigned long long> host((size_t)splitK * 16);
            bool ok = hipMalloc(&A, span(op.A.desc)) == hipSuccess && hipMalloc(&B, span(op.B.desc)) == hipSuccess &&
                      hipMalloc(&D, span(op.D.desc)) == hipSuccess && hipMalloc(&W, pl.requiredWorkspace) == hipSuccess &&
                      hipMalloc((void**)&T, tBytes) == hipSuccess;
            if (ok) {
                (void)hipMemset(A, 0x3c, span(op.A.desc));
                (void)hipMemset(B, 0x3c, span(op.B.desc));
                (void)hipMemset(T, 0, tBytes);
                GettParams gp = pl.gett;
                gp.A = pl.view.swapped ? B : A;
                gp.B = pl.view.swapped ? A : B;
                gp.endA += (unsigned long long)(uintptr_t)gp.A;
                gp.endB += (unsigned long long)(uintptr_t)gp.B;
                gp.C = D; gp.D = D; gp.alpha = 1.f; gp.beta = 0.f;
                gp.partial = static_cast<float*>(W);
                gp.sync = nullptr;
                gp.xcdTiles = 0;
                for (int rep = 0; rep < 4 && ok; ++rep) {
                    gp.timing = (rep == 3) ? T : nullptr;
                    ok = k.launch(gp, nullptr) == hipSuccess;
                }
                ok = ok && hipDeviceSynchronize() == hipSuccess && hipMemcpy(host.data(), T, tBytes, hipMemcpyDeviceToHost) == hipSuccess;
            }
            if (ok) {
                // workgroup b runs on XCD b % 8 (observed placement; only the weights depend on it)
                double clk[8] = {0}, n[8] = {0};
                for (uint32_t b = 0; b < splitK; ++b) {
                    const unsigned long long* t = &host[(size_t)b * 16];
                    const double cyc = (double)(t[4] - t[0]), wall = (double)(t[6] - t[5]);
                    if (t[4] > t[0] && t[6] > t[5]) { clk[b & 7] += cyc / wall; n[b & 7] += 1.0; }
                }
                double mean = 0.0;
                bool all = true;
                for (int x = 0; x < 8; ++x) { all = all && n[x] > 0; if (n[x] > 0) clk[x] /= n[x]; mean += clk[x] / 8.0; }
                if (all && mean > 0)
                    for (int x = 0; x < 8; ++x) {
                        const double w = clk[x] / mean;
                        handle->xcdSpeed[x] = (w > 0.9 && w < 1.1) ? w : 1.0;   // a die 10 % off is a measurement artefact
                    }
                CT_LOG("xcd calibration: relative clocks %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f", handle->xcdSpeed[0], handle->xcdSpeed[1],
                       handle->xcdSpeed[2], handle->xcdSpeed[3], handle->xcdSpeed[4], handle->xcdSpeed[5], handle->xcdSpeed[6], handle->xcdSpeed[7]);
            } else {
                (void)hipGetLastError();
            }
            if (A) (void)hipFree(A);
            if (B) (void)hipFree(B);
            if (D) (void)hipFree(D);
            if (W) (void)hipFree(W);
            if (T) (void)hipFree(T);
        }
    }
    // largest-remainder apportionment of perXcdSum tiles
    double sum = 0.0;
    for (double w : handle->xcdSpeed) sum += w;
    uint32_t n[8];
    double frac[8];
    uint64_t assigned = 0;
    for (int x = 0; x < 8; ++x) {
        const double ideal = (double)perXcdSum * handle->xcdSpeed[x] / sum;
        n[x] = (uint32_t)ideal;
        frac[x] = ideal - n[x];
        assigned += n[x];
    }
    while (assigned < perXcdSum) {
        int best = 0;
        for (int x = 1; x < 8; ++x) if (frac[x] > frac[best]) best = x;
        n[best] += 1; frac[best] = -1.0; assigned += 1;
    }
    unsigned long long packed = 0;
    bool uniform = true;
    for (int x = 0; x < 8; ++x) {
        if (n[x] == 0 || n[x] > 255) return 0;
        uniform = uniform && n[x] == n[0];
        packed |= (unsigned long long)n[x] << (8 * x);
    }
    return uniform ? 0 : packed;
}

// contraction.cu:218-222, elementwise_permute.cu:183-187 (limit 0), einsum.cu:324-329 (limit 1 GiB)
cutensorStatus_t cutensorCreatePlan(const cutensorHandle_t handle, cutensorPlan_t* plan,
                                    const cutensorOperationDescriptor_t desc, const cutensorPlanPreference_t pref,
                                    uint64_t workspaceSizeLimit) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || desc == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorPlanPreference defaults;
    const cutensorPlanPreference& pr = pref ? *pref : defaults;
    // Plan memo first (einsum.cu:264-329 plans inside every call with the cache on, :443-445): a repeated problem is one
    // hash + one lookup + one clone; ranking candidates, string keys and tile arithmetic happen on a miss only.
    PlanMemoKey mkey;
    uint64_t mhash = 0;
    const bool memoable = handle->planCacheCapacity > 0 && pr.cacheMode != CUTENSOR_CACHE_MODE_NONE && !plan_env_override() &&
                          build_memo_key(*desc, pr, workspaceSizeLimit, mkey);
    if (memoable) {
        mhash = mkey.hash();
        if (handle->pendingCount.load(std::memory_order_relaxed) > 0) resolve_pending_measurements(handle);
        std::shared_ptr<const cutensorPlan> proto;
        {
            std::lock_guard<std::mutex> g(handle->mtx);
            auto it = handle->planMemo.find(mhash);
            if (it != handle->planMemo.end() && it->second.key == mkey) {
                proto = it->second.proto;
                it->second.stamp = ++handle->memoClock;
            }
        }
        if (proto) {
            cutensorPlan* clone = clone_plan(*proto);
            if (clone == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
            handle->memoHits.fetch_add(1, std::memory_order_relaxed);
            *plan = clone;
            return CUTENSOR_STATUS_SUCCESS;
        }
        handle->memoMisses.fetch_add(1, std::memory_order_relaxed);
    }
    // the plan under construction is owned here until it is handed to the caller: an exception below (bad_alloc in a planner's vectors,
    // caught by the barrier at the end) or an early return frees it together with its sub-plans (round-5 advice)
    std::unique_ptr<cutensorPlan> owner(new (std::nothrow) cutensorPlan());
    cutensorPlan* const pl = owner.get();
    if (pl == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    pl->kind = desc->kind;
    pl->dtype = desc->A.desc.dtype;
    pl->scalarType = desc->scalarType;
    pl->alignA = desc->A.desc.alignment;
    pl->alignB = desc->B.present ? desc->B.desc.alignment : 0;
    pl->alignC = desc->C.present ? desc->C.desc.alignment : 0;
    pl->alignD = desc->D.desc.alignment;
    pl->accumulate64 = desc->compute && desc->compute->id == 5;
    std::string why;
    cutensorStatus_t st = CUTENSOR_STATUS_SUCCESS;

    if (desc->kind == OpKind::BlockSparseContraction) {
        st = blocksparse_plan(handle, *desc, workspaceSizeLimit, pl);
        if (st != CUTENSOR_STATUS_SUCCESS) { return st; }
        *plan = owner.release();
        return CUTENSOR_STATUS_SUCCESS;
    }
    if (desc->kind == OpKind::ContractionTrinary) {
        const uint64_t tOff = (desc->tBytes + 255) & ~255ull;
        if (workspaceSizeLimit < tOff) { CT_LOG("cutensorCreatePlan: trinary contraction needs %llu bytes for its intermediate", (unsigned long long)tOff); return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE; }
        cutensorOperationDescriptor s1 = desc->sub[0], s2 = desc->sub[1];
        cutensorPlan_t p1 = nullptr, p2 = nullptr;
        st = cutensorCreatePlan(handle, &p1, &s1, pref, workspaceSizeLimit - tOff);
        if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlan(handle, &p2, &s2, pref, workspaceSizeLimit - tOff);
        if (st != CUTENSOR_STATUS_SUCCESS) { delete p1; delete p2; return st; }
        pl->sub1 = p1; pl->sub2 = p2;
        pl->tBytes = desc->tBytes;
        for (int i = 0; i < 3; ++i) pl->triOrder[i] = desc->triOrder[i];
        pl->alignB3 = desc->B.desc.alignment;            // B
        pl->alignC = desc->C.desc.alignment;             // C (third input)
        pl->alignD = desc->E.desc.alignment;             // output E (and its beta source D)
        pl->requiredWorkspace = tOff + std::max(p1->requiredWorkspace, p2->requiredWorkspace);
        *plan = owner.release();
        return CUTENSOR_STATUS_SUCCESS;
    }

    if (desc->kind == OpKind::Contraction) {
        LoneSplit ls;
        if (split_lone_modes(*desc, ls)) {
            // reduce the operand(s) over the modes nothing else carries into packed temporaries at the head of the workspace, then contract
            const uint64_t offB = (ls.bytesA + 255) & ~255ull, offW = offB + ((ls.bytesB + 255) & ~255ull);
            if (workspaceSizeLimit < offW) { CT_LOG("cutensorCreatePlan: a contraction with a mode that one input alone carries needs %llu bytes for its temporaries", (unsigned long long)offW); return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE; }
            cutensorPlan_t pi = nullptr, pa = nullptr, pb = nullptr;
            st = cutensorCreatePlan(handle, &pi, &ls.inner, pref, workspaceSizeLimit - offW);
            if (st == CUTENSOR_STATUS_SUCCESS && ls.hasA) st = cutensorCreatePlan(handle, &pa, &ls.redA, pref, workspaceSizeLimit - offW);
            if (st == CUTENSOR_STATUS_SUCCESS && ls.hasB) st = cutensorCreatePlan(handle, &pb, &ls.redB, pref, workspaceSizeLimit - offW);
            if (st != CUTENSOR_STATUS_SUCCESS) { delete pi; delete pa; delete pb; return st; }
            pl->sub1 = pi; pl->loneA = pa; pl->loneB = pb;
            pl->loneBytesA = ls.bytesA; pl->loneBytesB = ls.bytesB;
            pl->choice = ContractionChoice{};
            pl->choice.kernel = -4;
            pl->requiredWorkspace = offW + std::max<uint64_t>(pi->requiredWorkspace, std::max<uint64_t>(pa ? pa->requiredWorkspace : 0, pb ? pb->requiredWorkspace : 0));
            CT_LOG("plan: contraction with modes that one input alone carries -> %s%sreduced first (%llu + %llu bytes of temporaries), then the contraction",
                   pa ? "A " : "", pb ? "B " : "", (unsigned long long)ls.bytesA, (unsigned long long)ls.bytesB);
            if (memoable) memo_insert(handle, mkey, mhash, *pl);
            *plan = owner.release();
            return CUTENSOR_STATUS_SUCCESS;
        }
        st = build_contraction_view(*desc, pl->view, &why);
        if (st != CUTENSOR_STATUS_SUCCESS) { return st; }
        if (pl->view.wide && !(CTAMD_HOOK_ENV("CUTENSOR_AMD_PEEL") && CTAMD_HOOK_ENV("CUTENSOR_AMD_PEEL")[0] == '0')) {
            // too many unfusable modes in a group for the tiled kernels: peel the smallest ones into a host loop if that takes
            // at most kMaxPeelLaunches launches (peel_wide_contraction), else fall through to the mode-table kernel
            cutensorOperationDescriptor inner;
            std::vector<PeelMode> peel;
            if (peel_wide_contraction(*desc, inner, peel)) {
                peel_fix_alignment(inner, peel);
                cutensorPlan_t ip = nullptr;
                if (cutensorCreatePlan(handle, &ip, &inner, pref, workspaceSizeLimit) == CUTENSOR_STATUS_SUCCESS && ip->choice.kernel != -2 &&
                    ip->sub1 == nullptr) {
                    // A peeled CONTRACTED mode accumulates into D: every launch after the first reads D as its C operand.  The inner
                    // plan carries the caller's C layout (and its conjugation); when that differs from D's, the accumulate launches
                    // get a second inner plan whose C descriptor is D's.
                    bool contracted = false;
                    for (const PeelMode& pm : peel) contracted = contracted || pm.contracted;
                    if (contracted && (inner.C.desc.stride != inner.D.desc.stride || inner.C.op != CUTENSOR_OP_IDENTITY)) {
                        cutensorOperationDescriptor inner2 = inner;
                        inner2.C = inner.D;
                        inner2.C.op = CUTENSOR_OP_IDENTITY;
                        cutensorPlan_t ip2 = nullptr;
                        if (cutensorCreatePlan(handle, &ip2, &inner2, pref, workspaceSizeLimit) != CUTENSOR_STATUS_SUCCESS || ip2->choice.kernel == -2 ||
                            ip2->sub1 != nullptr) {
                            delete ip2;
                            delete ip;
                            
                            return CUTENSOR_STATUS_NOT_SUPPORTED;
                        }
                        pl->sub2 = ip2;
                        ip->requiredWorkspace = std::max(ip->requiredWorkspace, ip2->requiredWorkspace);
                    }
                    pl->sub1 = ip;
                    pl->peel = peel;
                    pl->choice = ContractionChoice{};
                    pl->choice.kernel = -3;
                    pl->requiredWorkspace = ip->requiredWorkspace;
                    int64_t launches = 1;
                    for (const PeelMode& pm : peel) launches *= pm.extent;
                    CT_LOG("plan: contraction with an oversized mode group -> %zu mode(s) peeled, %lld launches of the tiled inner plan", peel.size(), (long long)launches);
                    *plan = owner.release();
                    return CUTENSOR_STATUS_SUCCESS;
                }
                delete ip;
            }
        }
        {
            const bool cplx = pl->view.dtype == HIP_C_32F || pl->view.dtype == HIP_C_64F;
            const bool genOff = CTAMD_HOOK_ENV("CUTENSOR_AMD_GEN") && CTAMD_HOOK_ENV("CUTENSOR_AMD_GEN")[0] == '0';
            // complex data only multiplies on the general MFMA family or on the mode-table kernel (complex scalars of the data's type)
            if (cplx && (genOff || desc->scalarType != pl->view.dtype || (pl->view.dtype == HIP_C_32F && pl->accumulate64))) pl->view.wide = true;
        }
        if (pl->view.wide) {
            // mode-table kernel: output modes (L, M, N), then contracted modes, in device memory owned by the plan
            const ContractionView& v = pl->view;
            std::vector<WideMode> tab;
            uint64_t outTotal = 1;
            for (const std::vector<CanonMode>* g : {&v.L, &v.M, &v.N})
                for (const CanonMode& m : *g) {
                    tab.push_back(WideMode{make_fastdiv((uint32_t)m.extent), m.sA, m.sB, m.sC, m.sD});
                    outTotal *= (uint64_t)m.extent;
                }
            pl->wide.nOut = (uint32_t)tab.size();
            for (const CanonMode& m : v.K) tab.push_back(WideMode{make_fastdiv((uint32_t)m.extent), m.sA, m.sB, 0, 0});
            pl->wide.nK = (uint32_t)v.K.size();
            pl->wide.outTotal = outTotal;
            pl->wide.kTotal = (uint32_t)v.totK;
            // conjugation flags follow the operands into their kernel roles (kernel-A is the user's B when swapped)
            const bool cA = desc->A.op == CUTENSOR_OP_CONJ, cB = desc->B.op == CUTENSOR_OP_CONJ;
            pl->wide.conjA = v.swapped ? cB : cA;
            pl->wide.conjB = v.swapped ? cA : cB;
            pl->wide.conjC = desc->C.op == CUTENSOR_OP_CONJ;
            // the table goes to device memory with the first cutensorContract: planning needs no GPU (and a plan that is never
            // executed allocates nothing)
            pl->wideTab = tab;
            if (pl->wideTab.empty()) pl->wideTab.push_back(WideMode{make_fastdiv(1), 0, 0, 0, 0});
            // The table goes to device memory HERE when the handle has a device — on the handle's device, outside any stream
            // capture, so that cutensorContract neither allocates nor synchronises (it may be called while a graph is being
            // captured).  Handles without a device (plan-only, the CPU tests) keep the host copy; a plan that one of those hands
            // to a process with a GPU uploads at first execution.
            if (handle->haveDevice) {
                int prev = -1;
                void* dev = nullptr;
                const size_t bytes = pl->wideTab.size() * sizeof(WideMode);
                const bool switched = hipGetDevice(&prev) == hipSuccess && prev != handle->device && hipSetDevice(handle->device) == hipSuccess;
                if (hipMalloc(&dev, bytes) == hipSuccess && hipMemcpy(dev, pl->wideTab.data(), bytes, hipMemcpyHostToDevice) == hipSuccess)
                    pl->wide.modes = static_cast<const WideMode*>(dev);
                else {
                    (void)hipGetLastError();
                    if (dev) (void)hipFree(dev);
                }
                if (switched) (void)hipSetDevice(prev);
            }
            pl->choice = ContractionChoice{};
            pl->choice.kernel = -2;
            pl->requiredWorkspace = 0;
            CT_LOG("plan: contraction with %u output + %u contracted unfusable modes -> mode-table kernel", pl->wide.nOut, pl->wide.nK);
            *plan = owner.release();
            return CUTENSOR_STATUS_SUCCESS;
        }
        const bool mfmaPath = pl->view.dtype == HIP_R_32F && !pl->accumulate64;
        const bool h16Path = !mfmaPath && !pl->accumulate64 && desc->scalarType == HIP_R_32F &&
                             (pl->view.dtype == HIP_R_16BF || pl->view.dtype == HIP_R_16F);
        // general MFMA family: 16-bit shapes the aligned kernels refuse, fp64 (double scalars), complex (complex scalars)
        const bool genPath = h16Path || (pl->view.dtype == HIP_R_64F && desc->scalarType == HIP_R_64F) ||
                             (pl->view.dtype == HIP_C_32F && desc->scalarType == HIP_C_32F && !pl->accumulate64) ||
                             (pl->view.dtype == HIP_C_64F && desc->scalarType == HIP_C_64F);
        ContractionChoice pick;   // kernel = -1: simple kernel
        std::vector<ContractionChoice> ch;
        if (mfmaPath) ch = rank_contraction_choices(pl->view, workspaceSizeLimit, handle->numCUs, pr.operandsStreamed != 0);
        else if (h16Path && !(CTAMD_HOOK_ENV("CUTENSOR_AMD_GEN") && CTAMD_HOOK_ENV("CUTENSOR_AMD_GEN")[0] == 'f'))   // "force" (measurement): the general family also where the aligned 16-bit kernels apply
            ch = rank_h16_choices(pl->view, workspaceSizeLimit, handle->numCUs);
        double tDirect32 = (mfmaPath && desc->scalarType == HIP_R_32F && (int)pr.algo < 0 && pr.kernelRank == 0 && !ctamd_research_env("CUTENSOR_AMD_KORDER"))
                               ? f32_direct_estimate_us(pl->view, ch) : 0.0;
        if ((pl->view.dtype == HIP_R_64F || pl->view.dtype == HIP_C_32F) && desc->scalarType == pl->view.dtype && genPath && ch.empty()) {   // fp64 / complex64 on element gathers (plan_repack)
            ContractionChoice g64;
            if (pick_gen_choice(pl->view, workspaceSizeLimit, handle->numCUs, g64)) tDirect32 = f64_direct_estimate_us(pl->view, g64);
        }
        if (((h16Path && (ch.empty() || h16_sweep_waste(pl->view)) && !CTAMD_HOOK_ENV("CUTENSOR_AMD_GEN") && !CTAMD_HOOK_ENV("CUTENSOR_AMD_H16_WAVES")) || tDirect32 > 0.0) &&
            (int)pr.algo < 0 && pr.kernelRank == 0 &&                       // (a caller who names a candidate gets that candidate)
            !(CTAMD_HOOK_ENV("CUTENSOR_AMD_REPACK") && CTAMD_HOOK_ENV("CUTENSOR_AMD_REPACK")[0] == '0')) {
            // the LDS-DMA kernels refuse the operands as they lie (or would spend most of every K-tile on the padding of a short ragged
            // contracted mode): copy them into packed temporaries first when that pays (plan_repack)
            RepackSplit rs;
            if (plan_repack(handle, *desc, pl->view, workspaceSizeLimit, tDirect32 > 0.0 ? tDirect32 : ch.empty() ? -1.0 : ch[0].estimateUs, rs)) {
                const uint64_t offB = (rs.bytesA + 255) & ~255ull, offW = offB + ((rs.bytesB + 255) & ~255ull);
                cutensorPlan_t pi = nullptr, pa = nullptr, pb = nullptr;
                {
                    RepackScope scope;
                    st = cutensorCreatePlan(handle, &pi, &rs.inner, pref, workspaceSizeLimit - offW);
                }
                if (st == CUTENSOR_STATUS_SUCCESS && rs.hasA) st = cutensorCreatePlan(handle, &pa, &rs.permA, pref, 0);
                if (st == CUTENSOR_STATUS_SUCCESS && rs.hasB) st = cutensorCreatePlan(handle, &pb, &rs.permB, pref, 0);
                if (st == CUTENSOR_STATUS_SUCCESS && pi->choice.family == (mfmaPath ? 0 : h16Path ? 1 : 2) && pi->sub1 == nullptr) {
                    pl->sub1 = pi; pl->loneA = pa; pl->loneB = pb;
                    pl->loneBytesA = rs.bytesA; pl->loneBytesB = rs.bytesB;
                    pl->choice = ContractionChoice{};
                    pl->choice.kernel = -4;
                    pl->requiredWorkspace = offW + pi->requiredWorkspace;
                    CT_LOG("plan: 16-bit contraction whose operands the LDS-DMA kernels cannot stage -> %s%scopied into packed temporaries first (%llu + %llu bytes), then the contraction",
                           pa ? "A " : "", pb ? "B " : "", (unsigned long long)rs.bytesA, (unsigned long long)rs.bytesB);
                    if (memoable && !t_inRepack) memo_insert(handle, mkey, mhash, *pl);
                    *plan = owner.release();
                    return CUTENSOR_STATUS_SUCCESS;
                }
                delete pi; delete pa; delete pb;                            // (the copies or the inner plan refused: the general family takes the problem as it is)
                st = CUTENSOR_STATUS_SUCCESS;
            }
        }
        if (ch.empty() && genPath && !(CTAMD_HOOK_ENV("CUTENSOR_AMD_GEN") && CTAMD_HOOK_ENV("CUTENSOR_AMD_GEN")[0] == '0')) {
            ContractionChoice g;
            if (pick_gen_choice(pl->view, workspaceSizeLimit, handle->numCUs, g)) ch.push_back(g);
        }
        if (!ch.empty()) {
            size_t idx = 0;
            // (a plan made under the "operands are streamed" preference neither reads nor feeds the per-problem cache: the cache is keyed by
            // the problem alone, and its entry belongs to the default policy)
            const bool useCache = handle->planCacheCapacity > 0 && pr.cacheMode != CUTENSOR_CACHE_MODE_NONE && pr.operandsStreamed == 0;
            const bool explicitPick = (int)pr.algo >= 0 || pr.kernelRank > 0;   // the caller names a candidate: the cache has no say
            const bool incremental = useCache && !explicitPick && pr.autotune == CUTENSOR_AUTOTUNE_MODE_INCREMENTAL;
            const bool patient = pr.algo == CUTENSOR_ALGO_DEFAULT_PATIENT;
            bool needKey = incremental || (useCache && patient);
            if (useCache && !explicitPick && !needKey) {
                std::lock_guard<std::mutex> g(handle->mtx);
                needKey = !handle->planCache.empty();
            }
            const std::string key = needKey ? problem_key(*desc) : std::string();   // the string form is what the cache FILE holds
            bool decided = false;
            // incremental autotuning (contraction_plan_cache.cu:215-237): the first INCREMENTAL_COUNT plans of a problem
            // are trials of candidates 0, 1, ... (timed by cutensorContract: the best of a trial plan's first few executions);
            // after that — and for every plan without the autotune mode — the cache answers with the fastest candidate
            // measured so far.  (The sample's loop is count + 1 rounds of which the last must hit the cache, :262.)
            if (incremental) {
                resolve_pending_measurements(handle);
                std::lock_guard<std::mutex> g(handle->mtx);
                auto tit = handle->tuning.find(key);
                if (tit == handle->tuning.end() && handle->tuning.size() < std::max<size_t>(handle->planCacheCapacity, 1))
                    tit = handle->tuning.emplace(key, cutensorHandle::TuneState{}).first;   // bounded like the cache itself
                if (tit != handle->tuning.end()) {
                    cutensorHandle::TuneState& t = tit->second;
                    const int limit = std::min<int>(std::max<int32_t>(pr.incrementalCount, 1), (int)ch.size());
                    if (t.next < limit) {
                        idx = (size_t)t.next++;
                        pl->tuneKey = key;
                        decided = true;
                    }
                }
            }
            if (!decided && useCache && !explicitPick) {
                std::lock_guard<std::mutex> g(handle->mtx);
                auto it = needKey ? handle->planCache.find(key) : handle->planCache.end();
                if (it != handle->planCache.end())
                    for (size_t i = 0; i < ch.size(); ++i)
                        if (ch[i].kernel == it->second.kernel && ch[i].splitK == it->second.splitK) { idx = i; decided = true; break; }
            }
            if (!decided) {
                if ((int)pr.algo >= 0) idx = std::min<size_t>((size_t)pr.algo, ch.size() - 1);
                else if (pr.kernelRank > 0) idx = std::min<size_t>((size_t)pr.kernelRank, ch.size() - 1);
                else if (patient) idx = (size_t)autotune_contraction(handle, *desc, pl->view, ch);
                if (const char* f = ctamd_research_env("CUTENSOR_AMD_FORCE")) {   // "kernel:splitK" experiment knob
                    int fk = -1; unsigned fs = 1;
                    if (std::sscanf(f, "%d:%u", &fk, &fs) >= 1)
                        for (size_t i = 0; i < ch.size(); ++i)
                            if (ch[i].kernel == fk && ch[i].splitK == fs) { idx = i; break; }
                }
                if (useCache && patient) {
                    std::lock_guard<std::mutex> g(handle->mtx);
                    if (handle->planCache.size() < handle->planCacheCapacity) {
                        handle->planCache[key] = PlanCacheEntry{key, ch[idx].kernel, ch[idx].splitK};
                        handle->planMemo.clear();   // a DEFAULT prototype memoised earlier must not shadow the measured choice
                    }
                }
            }
            pick = ch[idx];
        }
        pl->choice = pick;
        fill_gett_params(pl->view, pick, pl->gett, pl->skr);
        {   // conjugation follows the operands into their kernel roles (kernel-A is the user's B when swapped)
            const bool cA = desc->A.op == CUTENSOR_OP_CONJ, cB = desc->B.op == CUTENSOR_OP_CONJ;
            pl->gett.conjA = pl->view.swapped ? cB : cA;
            pl->gett.conjB = pl->view.swapped ? cA : cB;
            pl->gett.conjC = desc->C.op == CUTENSOR_OP_CONJ;
        }
        pl->requiredWorkspace = pick.workspace;
        // Per-XCD K split (one output tile, split-K over whole XCD rows): see calibrate_xcd_split
        if (mfmaPath && pick.kernel >= 0 && pick.family == 0 && pick.splitK >= 64 && pl->view.totL == 1 && pl->gett.tilesM * pl->gett.tilesN == 1) {
            int cnt = 0;
            const GettKernelInfo* tabf = gett_f32_kernels(&cnt);
            const char* env = ctamd_research_env("CUTENSOR_AMD_XCD_BALANCE");
            // opt-in: on the parts measured the clocks differ by +-1.5 %, below the 1-tile-in-32 (3 %) granularity
            // of the headline split, so the apportionment comes out uniform
            if (env && env[0] == '1' && tabf[pick.kernel].fragPartials && !tabf[pick.kernel].ablation)
                pl->gett.xcdTiles = calibrate_xcd_split(handle, *desc, *pl);
        }
        // In-launch fold of the split-K partials: only when every workgroup of the launch owns a CU of its own
        // (they wait for each other) and the output is a plain matrix; otherwise the fold is a second kernel.
        if (mfmaPath && pick.kernel >= 0 && pick.family == 0 && pick.splitK > 1) {
            int cnt = 0;
            const GettKernelInfo* tabf = gett_f32_kernels(&cnt);
            const char* env = CTAMD_HOOK_ENV("CUTENSOR_AMD_FUSED_FOLD");
            const bool allowed = env && env[0] == '1';   // opt-in: measured slower than the two-kernel fold
            if (allowed && tabf[pick.kernel].fragPartials && !tabf[pick.kernel].ablation && pl->gett.nBlocks <= (uint32_t)handle->numCUs &&
                pl->view.totL == 1 && pl->view.M.size() <= 1 && pl->view.N.size() <= 1) {
                std::lock_guard<std::mutex> g(handle->mtx);
                if (handle->syncPool == nullptr) {
                    void* ptr = nullptr;
                    const size_t bytes = (size_t)cutensorHandle::kSyncSlots * 64;
                    if (hipMalloc(&ptr, bytes) == hipSuccess && hipMemset(ptr, 0, bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess)
                        handle->syncPool = static_cast<uint32_t*>(ptr);
                    else
                        (void)hipGetLastError();
                }
                pl->fusedFold = handle->syncPool != nullptr;
            }
        }
        pl->requiredWorkspace = pick.workspace;
        if (log_level() > 0) {
            int count = 0;
            const GettKernelInfo* tab = gett_f32_kernels(&count);
            if (pick.family == 2) {
                int gc = 0;
                const GettKernelInfo* gt = gett_gen_kernels(&gc);
                CT_LOG("plan: contraction (general MFMA family) dtype=%d L=%llu M=%llu N=%llu K=%llu swapped=%d -> gen kernel %d (%dx%dx%d orientA=%d orientB=%d V=%d) splitK=%u",
                       (int)pl->view.dtype, (unsigned long long)pl->view.totL, (unsigned long long)pl->view.totM, (unsigned long long)pl->view.totN,
                       (unsigned long long)pl->view.totK, (int)pl->view.swapped, pick.kernel, gt[pick.kernel].bm, gt[pick.kernel].bn, gt[pick.kernel].bk,
                       gt[pick.kernel].layA, gt[pick.kernel].layB, gt[pick.kernel].vec, pick.splitK);
            } else if (pick.family == 1)
                CT_LOG("plan: contraction (16-bit MFMA) L=%llu M=%llu N=%llu K=%llu layA=%d layB=%d swapped=%d -> h16 kernel %d",
                       (unsigned long long)pl->view.totL, (unsigned long long)pl->view.totM, (unsigned long long)pl->view.totN,
                       (unsigned long long)pl->view.totK, pl->view.layA, pl->view.layB, (int)pl->view.swapped, pick.kernel);
            else if (pick.kernel >= 0)
                CT_LOG("plan: contraction L=%llu M=%llu N=%llu K=%llu layA=%d layB=%d swapped=%d -> kernel %d (%dx%dx%d) splitK=%u ws=%llu est=%.1fus",
                       (unsigned long long)pl->view.totL, (unsigned long long)pl->view.totM, (unsigned long long)pl->view.totN,
                       (unsigned long long)pl->view.totK, pl->view.layA, pl->view.layB, (int)pl->view.swapped, pick.kernel,
                       tab[pick.kernel].bm, tab[pick.kernel].bn, tab[pick.kernel].bk, pick.splitK,
                       (unsigned long long)pick.workspace, pick.estimateUs);
            else
                CT_LOG("plan: contraction -> simple kernel (dtype %d)", (int)pl->view.dtype);
        }
    } else if (desc->kind == OpKind::Reduction) {
        st = plan_reduction(*desc, workspaceSizeLimit, handle->numCUs, pl->red, &why);
        if (st != CUTENSOR_STATUS_SUCCESS) { return st; }
        pl->requiredWorkspace = pl->red.workspace;
        CT_LOG("plan: reduction variant=%d kept=%u red=%u splitR=%u perm=%d", pl->red.variant, pl->red.p.kept.total,
               pl->red.p.red.total, pl->red.p.splitR, (int)pl->red.isPermutation);
    } else if (desc->kind == OpKind::ElementwiseTrinary) {
        st = plan_elementwise_trinary(*desc, pl->ew3, &why);
        if (st != CUTENSOR_STATUS_SUCCESS) { return st; }
        pl->alignB3 = desc->B.desc.alignment;
        pl->requiredWorkspace = 0;
        CT_LOG("plan: elementwise trinary passes=%d swapAB=%d bothPermuted=%d variant(last)=%d", pl->ew3.twoPass ? 2 : 1, (int)pl->ew3.swapAB,
               (int)pl->ew3.bothPermuted, pl->ew3.last.variant);
    } else if (desc->kind == OpKind::Permutation && (!desc->padLeft.empty() || !desc->padRight.empty())) {
        // The output buffer is the packed tensor of extents e + padLeft + padRight (the sample sizes it that way,
        // elementwise_permute_padding.cu:101-103); the descriptor carries the unpadded extents.
        const size_t nm = desc->D.modes.size();
        std::vector<int64_t> packed(nm), padded(nm);
        int64_t accU = 1, accP = 1, offset = 0;
        bool isPacked = true;
        for (size_t i = 0; i < nm; ++i) {
            const int64_t l = desc->padLeft.empty() ? 0 : desc->padLeft[i], r = desc->padRight.empty() ? 0 : desc->padRight[i];
            packed[i] = accU; padded[i] = accP;
            isPacked = isPacked && (desc->D.desc.extent[i] == 1 || desc->D.desc.stride[i] == accU);
            offset += l * accP;
            accU *= desc->D.desc.extent[i];
            accP *= desc->D.desc.extent[i] + l + r;
        }
        if (!isPacked) { CT_LOG("cutensorCreatePlan: padding needs a packed output descriptor"); return CUTENSOR_STATUS_NOT_SUPPORTED; }
        cutensorOperationDescriptor inner = *desc;
        inner.D.desc.stride = padded;
        // the interior starts `offset` elements into the buffer: keep the 16-byte-lane variants only if that is lane-aligned
        const int64_t lane = 16 / (int64_t)dtype_size(desc->D.desc.dtype);
        if (offset % lane != 0) inner.D.desc.alignment = (uint32_t)dtype_size(desc->D.desc.dtype);
        st = plan_elementwise(inner, pl->ew, &why);
        if (st != CUTENSOR_STATUS_SUCCESS) { return st; }
        pl->padFillElems = (uint64_t)accP;
        pl->padOffsetElems = offset;
        pl->padValue = desc->padValue;
        pl->requiredWorkspace = 0;
        CT_LOG("plan: padded permutation variant=%d fill=%llu elems offset=%lld", pl->ew.variant, (unsigned long long)pl->padFillElems, (long long)offset);
    } else {
        st = plan_elementwise(*desc, pl->ew, &why);
        if (st != CUTENSOR_STATUS_SUCCESS) { return st; }
        pl->requiredWorkspace = 0;
        CT_LOG("plan: elementwise variant=%d E0=%u E1=%u rest=%u blocks=%u", pl->ew.variant, pl->ew.p.E0, pl->ew.p.E1,
               pl->ew.p.rest.total, pl->ew.p.nBlocks);
    }
    if (memoable) memo_insert(handle, mkey, mhash, *pl);
    *plan = owner.release();
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

cutensorPlan::~cutensorPlan() {
    delete sub1;
    delete sub2;
    delete loneA;
    delete loneB;
    if (wide.modes != nullptr) (void)hipFree(const_cast<ctamd::WideMode*>(wide.modes));
}

cutensorStatus_t cutensorDestroyPlan(cutensorPlan_t plan) try {
    delete plan;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// contraction.cu:231-235
cutensorStatus_t cutensorPlanGetAttribute(const cutensorHandle_t handle, const cutensorPlan_t plan,
                                          cutensorPlanAttribute_t attr, void* buf, size_t sizeInBytes) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || buf == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (attr == CUTENSOR_PLAN_REQUIRED_WORKSPACE && sizeInBytes == sizeof(uint64_t)) {
        *static_cast<uint64_t*>(buf) = plan->requiredWorkspace;
        return CUTENSOR_STATUS_SUCCESS;
    }
    return CUTENSOR_STATUS_INVALID_VALUE;
} CTAMD_API_CATCH

// ---- execution -----------------------------------------------------------------------------------
// contraction.cu:261-265, einsum.cu:334-338
cutensorStatus_t cutensorContract(const cutensorHandle_t handle, const cutensorPlan_t plan, const void* alpha,
                                  const void* A, const void* B, const void* beta, const void* C, void* D,
                                  void* workspace, uint64_t workspaceSize, cudaStream_t stream) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || plan->kind != OpKind::Contraction) return CUTENSOR_STATUS_INVALID_VALUE;
    if (alpha == nullptr || beta == nullptr || A == nullptr || B == nullptr || D == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    const double a = scalar_as_double(alpha, plan->scalarType), b = scalar_as_double(beta, plan->scalarType);
    const double aIm = scalar_imag(alpha, plan->scalarType), bIm = scalar_imag(beta, plan->scalarType);
    const bool betaZero = (b == 0.0 && bIm == 0.0);
    if (!betaZero && C == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (misaligned(A, plan->alignA) || misaligned(B, plan->alignB) || misaligned(D, plan->alignD) ||
        (!betaZero && misaligned(C, plan->alignC)))
        return CUTENSOR_STATUS_INVALID_VALUE;
    if (plan->requiredWorkspace > 0 && (workspace == nullptr || workspaceSize < plan->requiredWorkspace))
        return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE;

    if (plan->choice.kernel == -4) {
        // a mode that one input alone carries (split_lone_modes): reduce that input over it into its temporary, contract the temporaries
        if (plan->sub1 == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
        const uint64_t offB = (plan->loneBytesA + 255) & ~255ull, offW = offB + ((plan->loneBytesB + 255) & ~255ull);
        char* ws = static_cast<char*>(workspace);
        const float onef[2] = {1.f, 0.f}, zerof[2] = {0.f, 0.f};
        const double oned[2] = {1.0, 0.0}, zerod[2] = {0.0, 0.0};
        const bool wideScalar = plan->scalarType == HIP_R_64F || plan->scalarType == HIP_C_64F;
        const void* one = wideScalar ? static_cast<const void*>(oned) : static_cast<const void*>(onef);
        const void* zero = wideScalar ? static_cast<const void*>(zerod) : static_cast<const void*>(zerof);
        const void* a = A;
        const void* bb = B;
        cutensorStatus_t st = CUTENSOR_STATUS_SUCCESS;
        // (the first step of an operand is a reduction over its lone modes, or — plan_repack — a permuted copy into a packed temporary)
        if (plan->loneA) {
            st = plan->loneA->kind == OpKind::Permutation ? cutensorPermute(handle, plan->loneA, one, A, ws, stream)
                                                         : cutensorReduce(handle, plan->loneA, one, A, zero, ws, ws, ws + offW, workspaceSize - offW, stream);
            a = ws;
        }
        if (st == CUTENSOR_STATUS_SUCCESS && plan->loneB) {
            st = plan->loneB->kind == OpKind::Permutation ? cutensorPermute(handle, plan->loneB, one, B, ws + offB, stream)
                                                         : cutensorReduce(handle, plan->loneB, one, B, zero, ws + offB, ws + offB, ws + offW, workspaceSize - offW, stream);
            bb = ws + offB;
        }
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
        return cutensorContract(handle, plan->sub1, alpha, a, bb, beta, C, D, ws + offW, workspaceSize - offW, stream);
    }
    if (plan->choice.kernel == -3) {
        // peeled contraction: every index combination of the peeled modes is one launch of the inner plan on offset operands;
        // a combination whose contracted indices are all zero writes its region of D first (caller's beta, caller's C), the
        // others accumulate into it
        if (plan->sub1 == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
        const size_t es = dtype_size(plan->dtype);
        const float onef[2] = {1.f, 0.f};      // (real, imaginary): also the complex one
        const double oned[2] = {1.0, 0.0};
        const void* one = (plan->scalarType == HIP_R_64F || plan->scalarType == HIP_C_64F) ? static_cast<const void*>(oned) : static_cast<const void*>(onef);
        const size_t n = plan->peel.size();
        std::vector<int64_t> digit(n, 0);
        for (;;) {
            int64_t oA = 0, oB = 0, oC = 0, oD = 0;
            bool first = true;
            for (size_t i = 0; i < n; ++i) {
                const PeelMode& pm = plan->peel[i];
                oA += digit[i] * pm.sA; oB += digit[i] * pm.sB; oC += digit[i] * pm.sC; oD += digit[i] * pm.sD;
                if (pm.contracted && digit[i] != 0) first = false;
            }
            char* d = static_cast<char*>(D) + oD * (int64_t)es;
            const char* c = first ? (C ? static_cast<const char*>(C) + oC * (int64_t)es : nullptr) : d;
            // accumulate launches read D in D's own layout (sub2, when the caller's C is laid out differently or conjugated)
            const cutensorStatus_t st = cutensorContract(handle, (!first && plan->sub2) ? plan->sub2 : plan->sub1, alpha, static_cast<const char*>(A) + oA * (int64_t)es,
                                                         static_cast<const char*>(B) + oB * (int64_t)es, first ? beta : one, c, d, workspace,
                                                         workspaceSize, stream);
            if (st != CUTENSOR_STATUS_SUCCESS) return st;
            size_t i = 0;
            for (; i < n; ++i) {
                if (++digit[i] < plan->peel[i].extent) break;
                digit[i] = 0;
            }
            if (i == n) break;
        }
        return CUTENSOR_STATUS_SUCCESS;
    }

    GettParams p = plan->gett;
    p.A = plan->view.swapped ? B : A;
    p.B = plan->view.swapped ? A : B;
    p.C = (!betaZero) ? C : D;
    p.D = D;
    p.alpha = (float)a; p.beta = (float)b;
    p.alpha64 = a; p.beta64 = b;
    p.alphaIm = aIm; p.betaIm = bIm;
    p.endA += (unsigned long long)(uintptr_t)p.A;   // the plan holds the operands' byte spans (fill_gett_params)
    p.endB += (unsigned long long)(uintptr_t)p.B;
    p.timing = handle->timingBuffer.load(std::memory_order_relaxed);
    // incremental-autotuning trial: one event pair around everything this call launches, read later (resolve_pending_measurements)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // only the first kTrialTimedRuns executions of a trial plan are timed: a plan created once and run in a loop neither
    // grows the pending list nor pays event creation per launch
    if (!plan->tuneKey.empty() && plan->trial.timed.load(std::memory_order_relaxed) < cutensorPlan::kTrialTimedRuns &&
        plan->trial.timed.fetch_add(1, std::memory_order_relaxed) < cutensorPlan::kTrialTimedRuns) {
        if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) {
            (void)hipGetLastError();
            if (t0) (void)hipEventDestroy(t0);
            t0 = t1 = nullptr;
        } else {
            (void)hipEventRecord(t0, stream);
        }
    }
    hipError_t err;
    if (plan->choice.kernel == -2) {
        if (plan->wide.modes == nullptr) {   // first execution: the mode table moves to the device (kept until the plan dies)
            std::lock_guard<std::mutex> g(handle->mtx);
            if (plan->wide.modes == nullptr) {
                void* dev = nullptr;
                const size_t bytes = plan->wideTab.size() * sizeof(WideMode);
                if (hipMalloc(&dev, bytes) != hipSuccess || hipMemcpy(dev, plan->wideTab.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
                    (void)hipGetLastError();
                    if (dev) (void)hipFree(dev);
                    return CUTENSOR_STATUS_ALLOC_FAILED;
                }
                plan->wide.modes = static_cast<const WideMode*>(dev);
            }
        }
        WideParams w = plan->wide;
        w.A = p.A; w.B = p.B; w.C = p.C; w.D = D;
        w.alpha = p.alpha; w.beta = p.beta; w.alpha64 = a; w.beta64 = b;
        w.alphaIm = aIm; w.betaIm = bIm;
        err = launch_gett_wide(w, (int)plan->dtype, plan->accumulate64, stream);
        g_launchCounts[1].fetch_add(1, std::memory_order_relaxed);
    } else if (plan->choice.kernel < 0) {
        g_launchCounts[0].fetch_add(1, std::memory_order_relaxed);
        p.partial = nullptr;
        err = launch_gett_simple(p, (int)plan->dtype, plan->accumulate64, stream);
    } else if (plan->choice.family == 1 || plan->choice.family == 2) {
        int count = 0;
        const GettKernelInfo* tab = plan->choice.family == 2 ? gett_gen_kernels(&count) : gett_h16_kernels(&count);
        g_launchCounts[plan->choice.family == 2 ? 4 : 3].fetch_add(1, std::memory_order_relaxed);
        p.partial = (plan->choice.splitK > 1) ? static_cast<float*>(workspace) : nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (handle->prof.enabled.load(std::memory_order_relaxed) && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
            (void)hipEventRecord(e0, stream);
        int launchKernel = plan->choice.kernel;
        // beta is known only now.  The persistent 16-bit kernel (table entries 88..95) streams its tiles with beta != 0 too since round 6
        // (C joins the accumulators through the idle row image: gett_h16p.hip) — when C has the 16-byte lanes of D (its fastest N mode
        // contiguous).  Any other C sends every tile through the ring-resident epilogue, where the kernel is slower than its one-tile
        // twin (48..55: same tile, same arguments, same workspace; profiles/r05r_h16p_beta.jsonl): the twin is launched then
        // (CUTENSOR_AMD_H16_WAVES=4p names the kernel for every call: the tests of that path)
        static const bool persistentForced = [] { const char* e = CTAMD_HOOK_ENV("CUTENSOR_AMD_H16_WAVES"); return e && e[0] == '4' && e[1] == 'p'; }();
        if (plan->choice.family == 1 && tab[launchKernel].pf == 12 && b != 0.0 && p.cStrideN[0] != 1 && !persistentForced && launchKernel - 40 >= 0 &&
            tab[launchKernel - 40].pf == 7)
            launchKernel -= 40;
        if (plan->choice.family == 1) g_lastH16Kernel.store(launchKernel, std::memory_order_relaxed);
        if (plan->choice.family == 1 && plan->choice.stripKernel >= 0 && plan->choice.stripKernel < count) {
            // strip plan: the interior's whole tiles on the chosen kernel, then the two edge strips (rows past mInt x all columns, rows
            // below mInt x columns past nInt) as ONE launch of the 64 x 64 tile with two tile rectangles
            const GettKernelInfo& ki = tab[launchKernel];
            const GettKernelInfo& ks = tab[plan->choice.stripKernel];
            const uint32_t mInt = plan->choice.mInt, nInt = plan->choice.nInt, Mt = p.gM.total, Nt = p.gN.total, Lt = p.gL.total;
            GettParams q = p;
            q.tilesM = mInt / (uint32_t)ki.bm; q.tilesN = nInt / (uint32_t)ki.bn;
            q.nBlocks = q.tilesM * q.tilesN * Lt;
            err = ki.launch(q, stream);
            GettParams s2 = p;
            s2.mOrg = mInt; s2.nOrg = 0;
            s2.tilesM = (Mt - mInt + (uint32_t)ks.bm - 1u) / (uint32_t)ks.bm; s2.tilesN = (Nt + (uint32_t)ks.bn - 1u) / (uint32_t)ks.bn;
            s2.mOrg2 = 0; s2.nOrg2 = nInt;
            s2.tilesM2 = (mInt + (uint32_t)ks.bm - 1u) / (uint32_t)ks.bm; s2.tilesN2 = (Nt - nInt + (uint32_t)ks.bn - 1u) / (uint32_t)ks.bn;
            if (s2.tilesM * s2.tilesN == 0u) {     // no rows past the interior: the column strip is the only rectangle
                s2.mOrg = s2.mOrg2; s2.nOrg = s2.nOrg2; s2.tilesM = s2.tilesM2; s2.tilesN = s2.tilesN2;
                s2.tilesM2 = s2.tilesN2 = 0;
            }
            s2.nBlocks = (s2.tilesM * s2.tilesN + s2.tilesM2 * s2.tilesN2) * Lt;
            if (err == hipSuccess && s2.nBlocks > 0u) err = ks.launch(s2, stream);
        } else
        err = tab[launchKernel].launch(p, stream);
        if (e0 && e1) {
            (void)hipEventRecord(e1, stream);
            std::lock_guard<std::mutex> g(handle->prof.mtx);
            handle->prof.events.emplace_back(e0, e1);
        }
        if (err == hipSuccess && plan->choice.splitK > 1) {
            SplitKReduceParams r = plan->skr;
            r.partial = static_cast<float*>(workspace);
            r.C = p.C; r.D = D; r.alpha = p.alpha; r.beta = p.beta;
            r.alpha64 = a; r.beta64 = b; r.alphaIm = aIm; r.betaIm = bIm; r.conjC = p.conjC;
            const int elem = plan->choice.family == 2 ? tab[plan->choice.kernel].elem : GEN_BF16;
            err = (elem >= GEN_F64) ? launch_gen_splitk_reduce(r, elem, stream) : launch_splitk_reduce(r, stream);
        }
    } else {
        int count = 0;
        const GettKernelInfo* tab = gett_f32_kernels(&count);
        g_launchCounts[2].fetch_add(1, std::memory_order_relaxed);
        p.partial = (plan->choice.splitK > 1) ? static_cast<float*>(workspace) : nullptr;
        {
            static const int policy = [] { const char* e = CTAMD_HOOK_ENV("CUTENSOR_AMD_PARTIAL_STORE"); return e ? (e[0] == 'p' ? 1 : e[0] == 'n' ? 2 : e[0] == 's' ? 3 : 0) : 0; }();   // hooks flavour; 's': the row epilogue skips its stores (timing only)
            p.partialPolicy = policy;
        }
        if (plan->fusedFold) {
            uint32_t slot;
            {
                std::lock_guard<std::mutex> g(handle->mtx);
                slot = handle->syncNext++ % cutensorHandle::kSyncSlots;
            }
            p.sync = handle->syncPool + (size_t)slot * 16;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (handle->prof.enabled.load(std::memory_order_relaxed) && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
            (void)hipEventRecord(e0, stream);
        err = tab[plan->choice.kernel].launch(p, stream);
        if (e0 && e1) {
            (void)hipEventRecord(e1, stream);
            std::lock_guard<std::mutex> g(handle->prof.mtx);
            handle->prof.events.emplace_back(e0, e1);
        }
        if (err == hipSuccess && plan->choice.splitK > 1 && !plan->fusedFold && !handle->skipFold.load(std::memory_order_relaxed)) {
            SplitKReduceParams r = plan->skr;
            r.partial = static_cast<float*>(workspace);
            r.C = p.C; r.D = D; r.alpha = p.alpha; r.beta = p.beta;
            err = tab[plan->choice.kernel].fragPartials ? launch_splitk_reduce_frag(r, stream) : launch_splitk_reduce(r, stream);
        }
    }
    if (t0 != nullptr) {
        if (err == hipSuccess && hipEventRecord(t1, stream) == hipSuccess) {
            std::lock_guard<std::mutex> g(handle->mtx);
            handle->pending.push_back(cutensorHandle::PendingMeasurement{plan->tuneKey, plan->choice.kernel, plan->choice.splitK, t0, t1});
            handle->pendingCount.store((int)handle->pending.size(), std::memory_order_relaxed);
        } else {   // a failed launch is not a measurement
            (void)hipEventDestroy(t0);
            (void)hipEventDestroy(t1);
        }
    }
    if (err != hipSuccess) { CT_LOG("cutensorContract: %s", hipGetErrorString(err)); return CUTENSOR_STATUS_EXECUTION_FAILED; }
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

static hipError_t run_elementwise(const EwPlan& ew, hipDataType dtype, double a, const void* A, double g,
                                  const void* C, void* D, hipStream_t stream, const void* E = nullptr, double d = 0.0,
                                  double aIm = 0.0, double gIm = 0.0) {   // aIm / gIm: imaginary parts of alpha / gamma (complex data)
    Ew2DParams p = ew.p;
    p.A = A;
    // a zero gamma drops the C term only for ADD (alpha perm(A) + 0): MUL / MAX / MIN still need it
    p.C = (ew.usesC && (g != 0.0 || gIm != 0.0 || (p.opAC != 0 && p.opAC != CUTENSOR_OP_ADD))) ? C : nullptr;
    p.D = D;
    p.E = E;
    p.alpha = (float)a; p.gamma = (float)g; p.alpha64 = a; p.gamma64 = g;
    p.alphaIm = aIm; p.gammaIm = gIm;
    p.delta = (float)d; p.delta64 = d;
    return launch_elementwise(p, ew.variant, (int)dtype, stream);
}

// reduction.cu:219-222, einsum.cu:369-372
cutensorStatus_t cutensorReduce(const cutensorHandle_t handle, const cutensorPlan_t plan, const void* alpha,
                                const void* A, const void* beta, const void* C, void* D, void* workspace,
                                uint64_t workspaceSize, cudaStream_t stream) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || plan->kind != OpKind::Reduction) return CUTENSOR_STATUS_INVALID_VALUE;
    if (alpha == nullptr || beta == nullptr || A == nullptr || D == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    const double a = scalar_as_double(alpha, plan->scalarType), b = scalar_as_double(beta, plan->scalarType);
    const double aIm = scalar_imag(alpha, plan->scalarType), bIm = scalar_imag(beta, plan->scalarType);   // complex data: complex scalars
    const bool betaSet = b != 0.0 || bIm != 0.0;
    if (betaSet && C == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (misaligned(A, plan->alignA) || misaligned(D, plan->alignD) || (betaSet && misaligned(C, plan->alignC)))
        return CUTENSOR_STATUS_INVALID_VALUE;
    hipError_t err;
    if (plan->red.isPermutation) {
        err = run_elementwise(plan->red.perm, plan->dtype, a, A, b, C, D, stream, nullptr, 0.0, aIm, bIm);
    } else {
        if (plan->requiredWorkspace > 0 && (workspace == nullptr || workspaceSize < plan->requiredWorkspace))
            return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE;
        ReduceParams p = plan->red.p;
        p.A = A; p.C = betaSet ? C : D; p.D = D;
        p.alpha = (float)a; p.beta = (float)b; p.alpha64 = a; p.beta64 = b;
        p.alphaIm = aIm; p.betaIm = bIm;
        p.partial = (p.splitR > 1) ? workspace : nullptr;
        const bool acc64 = plan->accumulate64 || plan->dtype == HIP_R_64F;
        err = launch_reduce(p, plan->red.variant, (int)plan->dtype, acc64, stream);
        if (err == hipSuccess && p.splitR > 1) err = launch_reduce_finalize(p, (int)plan->dtype, acc64, stream);
    }
    if (err != hipSuccess) { CT_LOG("cutensorReduce: %s", hipGetErrorString(err)); return CUTENSOR_STATUS_EXECUTION_FAILED; }
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// elementwise_permute.cu:198-200
cutensorStatus_t cutensorPermute(const cutensorHandle_t handle, const cutensorPlan_t plan, const void* alpha,
                                 const void* A, void* B, const cudaStream_t stream) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || plan->kind != OpKind::Permutation) return CUTENSOR_STATUS_INVALID_VALUE;
    if (alpha == nullptr || A == nullptr || B == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (misaligned(A, plan->alignA) || misaligned(B, plan->alignD)) return CUTENSOR_STATUS_INVALID_VALUE;
    const double a = scalar_as_double(alpha, plan->scalarType);
    hipError_t err = hipSuccess;
    void* out = B;
    if (plan->padFillElems != 0) {   // border (and interior, rewritten next) = padding value
        err = launch_fill(B, plan->padFillElems, (int)plan->dtype, plan->padValue, stream);
        out = static_cast<char*>(B) + plan->padOffsetElems * (int64_t)dtype_size(plan->dtype);
    }
    if (err == hipSuccess) err = run_elementwise(plan->ew, plan->dtype, a, A, 0.0, nullptr, out, stream, nullptr, 0.0, scalar_imag(alpha, plan->scalarType));
    if (err != hipSuccess) { CT_LOG("cutensorPermute: %s", hipGetErrorString(err)); return CUTENSOR_STATUS_EXECUTION_FAILED; }
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// elementwise_binary.cu:202-205
cutensorStatus_t cutensorElementwiseBinaryExecute(const cutensorHandle_t handle, const cutensorPlan_t plan,
                                                  const void* alpha, const void* A, const void* gamma,
                                                  const void* C, void* D, cudaStream_t stream) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || plan->kind != OpKind::ElementwiseBinary) return CUTENSOR_STATUS_INVALID_VALUE;
    if (alpha == nullptr || gamma == nullptr || A == nullptr || C == nullptr || D == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (misaligned(A, plan->alignA) || misaligned(C, plan->alignC) || misaligned(D, plan->alignD)) return CUTENSOR_STATUS_INVALID_VALUE;
    const double a = scalar_as_double(alpha, plan->scalarType), g = scalar_as_double(gamma, plan->scalarType);
    hipError_t err = run_elementwise(plan->ew, plan->dtype, a, A, g, C, D, stream, nullptr, 0.0, scalar_imag(alpha, plan->scalarType),
                                     scalar_imag(gamma, plan->scalarType));
    if (err != hipSuccess) return CUTENSOR_STATUS_EXECUTION_FAILED;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// elementwise_trinary.cu:223-227
cutensorStatus_t cutensorElementwiseTrinaryExecute(const cutensorHandle_t handle, const cutensorPlan_t plan,
                                                   const void* alpha, const void* A, const void* beta, const void* B,
                                                   const void* gamma, const void* C, void* D, cudaStream_t stream) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || plan->kind != OpKind::ElementwiseTrinary) return CUTENSOR_STATUS_INVALID_VALUE;
    if (alpha == nullptr || beta == nullptr || gamma == nullptr || A == nullptr || B == nullptr || C == nullptr || D == nullptr)
        return CUTENSOR_STATUS_INVALID_VALUE;
    if (misaligned(A, plan->alignA) || misaligned(B, plan->alignB3) || misaligned(C, plan->alignC) || misaligned(D, plan->alignD))
        return CUTENSOR_STATUS_INVALID_VALUE;
    const double a = scalar_as_double(alpha, plan->scalarType), b = scalar_as_double(beta, plan->scalarType),
                 g = scalar_as_double(gamma, plan->scalarType);
    const EwTrinaryPlan& t = plan->ew3;
    hipError_t err = hipSuccess;
    if (t.bothPermuted) {
        Ew2DParams q = t.last.p;
        q.A = A; q.X = B; q.D = D; q.E = nullptr;
        q.C = (t.last.usesC && (g != 0.0 || (q.opAC != 0 && q.opAC != CUTENSOR_OP_ADD))) ? C : nullptr;
        q.alpha = (float)a; q.alpha64 = a; q.xi = (float)b; q.xi64 = b; q.gamma = (float)g; q.gamma64 = g;
        err = launch_elementwise(q, t.last.variant, (int)plan->dtype, stream);
    } else if (t.twoPass) {
        err = run_elementwise(t.first, plan->dtype, a, A, 0.0, nullptr, D, stream);                       // D = alpha perm(A)
        if (err == hipSuccess) err = run_elementwise(t.last, plan->dtype, b, B, g, C, D, stream, D, 1.0);  // combine in place
    } else if (t.swapAB) {
        err = run_elementwise(t.last, plan->dtype, a, A, g, C, D, stream, B, b);   // E = B (has D's layout)
    } else {
        err = run_elementwise(t.last, plan->dtype, b, B, g, C, D, stream, A, a);   // E = A
    }
    if (err != hipSuccess) { CT_LOG("cutensorElementwiseTrinaryExecute: %s", hipGetErrorString(err)); return CUTENSOR_STATUS_EXECUTION_FAILED; }
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// contraction_trinary.cu:290-294
cutensorStatus_t cutensorContractTrinary(const cutensorHandle_t handle, const cutensorPlan_t plan, const void* alpha,
                                         const void* A, const void* B, const void* C, const void* beta, const void* D, void* E,
                                         void* workspace, uint64_t workspaceSize, cudaStream_t stream) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || plan->kind != OpKind::ContractionTrinary || plan->sub1 == nullptr || plan->sub2 == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (alpha == nullptr || beta == nullptr || A == nullptr || B == nullptr || C == nullptr || E == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (workspace == nullptr || workspaceSize < plan->requiredWorkspace) return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE;
    if (misaligned(workspace, 256)) return CUTENSOR_STATUS_INVALID_VALUE;
    const void* in[3] = {A, B, C};
    const void *X = in[plan->triOrder[0]], *Y = in[plan->triOrder[1]], *Z = in[plan->triOrder[2]];
    const uint64_t tOff = (plan->tBytes + 255) & ~255ull;
    void* T = workspace;
    void* ws = static_cast<char*>(workspace) + tOff;
    // one / zero in the plan's scalar type; {re, im} pairs so that a complex scalar type reads a well-defined imaginary part
    const double one64[2] = {1.0, 0.0}, zero64[2] = {0.0, 0.0};
    const float one32[2] = {1.f, 0.f}, zero32[2] = {0.f, 0.f};
    const bool f64 = plan->scalarType == HIP_R_64F || plan->scalarType == HIP_C_64F;
    const void* one = f64 ? static_cast<const void*>(one64) : static_cast<const void*>(one32);
    const void* zero = f64 ? static_cast<const void*>(zero64) : static_cast<const void*>(zero32);
    cutensorStatus_t st = cutensorContract(handle, plan->sub1, one, X, Y, zero, T, T, ws, workspaceSize - tOff, stream);
    if (st != CUTENSOR_STATUS_SUCCESS) return st;
    return cutensorContract(handle, plan->sub2, alpha, T, Z, beta, D, E, ws, workspaceSize - tOff, stream);
} CTAMD_API_CATCH

// contraction_jit.cu:134,398 — the engine has no run-time code generation (every kernel is compiled ahead of
// time for gfx950), so its "kernel cache" holds nothing: writing produces a small tagged file, reading checks
// the tag and reports IO_ERROR for a missing file exactly as the sample expects on its first run.
cutensorStatus_t cutensorWriteKernelCacheToFile(const cutensorHandle_t handle, const char filename[]) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (filename == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    FILE* f = std::fopen(filename, "w");
    if (f == nullptr) return CUTENSOR_STATUS_IO_ERROR;
    std::fprintf(f, "cutensor-amd-kernelcache 1 gfx950 0\n");
    std::fclose(f);
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH
cutensorStatus_t cutensorReadKernelCacheFromFile(cutensorHandle_t handle, const char filename[]) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (filename == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    FILE* f = std::fopen(filename, "r");
    if (f == nullptr) return CUTENSOR_STATUS_IO_ERROR;
    char tag[64] = {0};
    const bool ok = std::fscanf(f, "%63s", tag) == 1 && std::strcmp(tag, "cutensor-amd-kernelcache") == 0;
    std::fclose(f);
    return ok ? CUTENSOR_STATUS_SUCCESS : CUTENSOR_STATUS_IO_ERROR;
} CTAMD_API_CATCH

// utils.cuh:38
const char* cutensorGetErrorString(const cutensorStatus_t error) {
    switch (error) {
        case CUTENSOR_STATUS_SUCCESS: return "CUTENSOR_STATUS_SUCCESS";
        case CUTENSOR_STATUS_NOT_INITIALIZED: return "CUTENSOR_STATUS_NOT_INITIALIZED";
        case CUTENSOR_STATUS_ALLOC_FAILED: return "CUTENSOR_STATUS_ALLOC_FAILED";
        case CUTENSOR_STATUS_INVALID_VALUE: return "CUTENSOR_STATUS_INVALID_VALUE";
        case CUTENSOR_STATUS_ARCH_MISMATCH: return "CUTENSOR_STATUS_ARCH_MISMATCH";
        case CUTENSOR_STATUS_MAPPING_ERROR: return "CUTENSOR_STATUS_MAPPING_ERROR";
        case CUTENSOR_STATUS_EXECUTION_FAILED: return "CUTENSOR_STATUS_EXECUTION_FAILED";
        case CUTENSOR_STATUS_INTERNAL_ERROR: return "CUTENSOR_STATUS_INTERNAL_ERROR";
        case CUTENSOR_STATUS_NOT_SUPPORTED: return "CUTENSOR_STATUS_NOT_SUPPORTED";
        case CUTENSOR_STATUS_LICENSE_ERROR: return "CUTENSOR_STATUS_LICENSE_ERROR";
        case CUTENSOR_STATUS_CUBLAS_ERROR: return "CUTENSOR_STATUS_CUBLAS_ERROR";
        case CUTENSOR_STATUS_CUDA_ERROR: return "CUTENSOR_STATUS_CUDA_ERROR";
        case CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE: return "CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE";
        case CUTENSOR_STATUS_INSUFFICIENT_DRIVER: return "CUTENSOR_STATUS_INSUFFICIENT_DRIVER";
        case CUTENSOR_STATUS_IO_ERROR: return "CUTENSOR_STATUS_IO_ERROR";
        default: return "<unknown>";
    }
}

size_t cutensorGetVersion(void) { return CUTENSOR_VERSION; }

// ---- diagnostics (not part of the cuTENSOR ABI; used by the tests and the bench) ----------------
// A contraction plan that fell to the mode-table kernel because a group has more unfusable modes than the tiled kernels'
// argument block describes: its canonical modes as (group 0 = L, 1 = M, 2 = N, 3 = K; caller's label; extent), at most maxOut of
// them; returns how many there are, 0 for every other plan.  cuTENSORMg uses it to peel one digit of an oversized group into a
// host loop (mg.cpp) instead of running the functional kernel.
int ctamdPlanModeTableGroups(const cutensorPlan_t plan, int32_t* group, int32_t* label, int64_t* extent, int maxOut) try {
    if (plan == nullptr || plan->kind != OpKind::Contraction || plan->choice.kernel != -2) return 0;
    if (plan->view.dtype == HIP_C_32F || plan->view.dtype == HIP_C_64F) return 0;     // complex data: not a matter of mode counts
    int n = 0;
    const std::vector<CanonMode>* gs[4] = {&plan->view.L, &plan->view.M, &plan->view.N, &plan->view.K};
    bool oversized = false;
    for (int g = 0; g < 4; ++g) oversized = oversized || (int)gs[g]->size() > kMaxGroupModes;
    if (!oversized) return 0;                                                          // wide for another reason (>= 2^31 elements)
    for (int g = 0; g < 4; ++g)
        for (const CanonMode& m : *gs[g]) {
            if (n < maxOut) { if (group) group[n] = g; if (label) label[n] = m.label; if (extent) extent[n] = m.extent; }
            ++n;
        }
    return n;
} CTAMD_API_CATCH_INT

// Launches of the inner plan a peeled contraction plan makes per call (peel_wide_contraction); 0 for every other plan.
int ctamdPlanPeelLaunches(const cutensorPlan_t plan) try {
    if (plan == nullptr || plan->kind != OpKind::Contraction || plan->choice.kernel != -3) return 0;
    long long n = 1;
    for (const PeelMode& pm : plan->peel) n *= pm.extent;
    return (int)n;
} CTAMD_API_CATCH_INT

// cutensorContract launches by kernel kind since the library was loaded: out[0] gett_simple_kernel (scalar FMA fallback), [1]
// gett_wide_kernel (mode table), [2] fp32 MFMA families, [3] aligned 16-bit MFMA family, [4] general MFMA family.  Lets a test that
// drives the library through someone else's binding (the reference's own einsum.cc) assert which kernels its cases ran on.
int ctamdLastH16Kernel(void) try { return g_lastH16Kernel.load(std::memory_order_relaxed); } CTAMD_API_CATCH_INT

void ctamdLaunchCounts(uint64_t out[5]) try {
    for (int i = 0; i < 5; ++i) out[i] = g_launchCounts[i].load(std::memory_order_relaxed);
} CTAMD_API_CATCH_VOID

// Plan-memo counters of this handle: plans answered by cloning a prototype / plans that went through the planner.
void ctamdPlanMemoStats(const cutensorHandle_t handle, uint64_t* hits, uint64_t* misses, uint32_t* entries) try {
    if (handle == nullptr) return;
    if (hits) *hits = handle->memoHits.load(std::memory_order_relaxed);
    if (misses) *misses = handle->memoMisses.load(std::memory_order_relaxed);
    if (entries) { std::lock_guard<std::mutex> g(handle->mtx); *entries = (uint32_t)handle->planMemo.size(); }
} CTAMD_API_CATCH_VOID
// Writes a one-line JSON description of the plan's kernel choice into buf.
int ctamdDescribePlan(const cutensorPlan_t plan, char* buf, size_t len) try {
    if (plan == nullptr || buf == nullptr || len == 0) return -1;
    int n = 0;
    if (plan->kind == OpKind::Contraction && plan->choice.kernel == -3 && plan->sub1 != nullptr) {
        // peeled contraction: the inner (tiled) plan's description with the peel in front
        long long launches = 1;
        for (const PeelMode& pm : plan->peel) launches *= pm.extent;
        n = std::snprintf(buf, len, "{\"peeled_modes\":%zu,\"peel_launches\":%lld,", plan->peel.size(), launches);
        if (n < 0 || (size_t)n >= len) return -1;
        const int m = ctamdDescribePlan(plan->sub1, buf + n - 1, len - (size_t)n + 1);   // overwrite our '{' + keep theirs: splice below
        if (m < 0) return -1;
        // buf now holds  {"peeled_modes":..,"peel_launches":..   followed (from n - 1) by the inner object  {...}: turn its '{' into ','
        buf[n - 1] = ',';
        return n - 1 + m;
    }
    if (plan->kind == OpKind::Contraction && plan->choice.kernel == -4 && plan->sub1 != nullptr) {
        // a mode that one input alone carries: which operands are reduced first, then the inner contraction's description
        const bool repA = plan->loneA && plan->loneA->kind == OpKind::Permutation, repB = plan->loneB && plan->loneB->kind == OpKind::Permutation;
        n = std::snprintf(buf, len, "{\"lone_reduce_A\":%d,\"lone_reduce_B\":%d,\"repack_A\":%d,\"repack_B\":%d,\"lone_bytes\":%llu,",
                          (plan->loneA && !repA) ? 1 : 0, (plan->loneB && !repB) ? 1 : 0, repA ? 1 : 0, repB ? 1 : 0,
                          (unsigned long long)(plan->loneBytesA + plan->loneBytesB));
        if (n < 0 || (size_t)n >= len) return -1;
        const int m = ctamdDescribePlan(plan->sub1, buf + n - 1, len - (size_t)n + 1);
        if (m < 0) return -1;
        buf[n - 1] = ',';
        return n - 1 + m;
    }
    if (plan->kind == OpKind::Contraction) {
        int count = 0;
        const GettKernelInfo* tab = (plan->choice.family == 2) ? gett_gen_kernels(&count) : (plan->choice.family == 1) ? gett_h16_kernels(&count) : gett_f32_kernels(&count);
        const int k = plan->choice.kernel;
        n = std::snprintf(buf, len,
                          "{\"op\":\"contraction\",\"family\":%d,\"L\":%llu,\"M\":%llu,\"N\":%llu,\"K\":%llu,\"swapped\":%d,\"layA\":%d,\"layB\":%d,"
                          "\"kernel\":%d,\"bm\":%d,\"bn\":%d,\"bk\":%d,\"wm\":%d,\"wn\":%d,\"wk\":%d,\"pf\":%d,\"abl\":%d,\"splitK\":%u,\"kPerSlice\":%u,"
                          "\"blocks\":%u,\"workspace\":%llu,\"model_us\":%.2f,\"fusedFold\":%d,\"xcdTiles\":\"%016llx\",\"kname\":\"%s\"",
                          plan->choice.family, (unsigned long long)plan->view.totL, (unsigned long long)plan->view.totM,
                          (unsigned long long)plan->view.totN, (unsigned long long)plan->view.totK, (int)plan->view.swapped,
                          plan->view.layA, plan->view.layB, k, k >= 0 ? tab[k].bm : 16, k >= 0 ? tab[k].bn : 16,
                          k >= 0 ? tab[k].bk : 16, k >= 0 ? tab[k].wm : 1, k >= 0 ? tab[k].wn : 1, k >= 0 ? tab[k].wk : 1,
                          k >= 0 ? tab[k].pf : 0, k >= 0 ? tab[k].ablation : 0,
                          plan->gett.splitK, plan->gett.kPerSlice, plan->gett.nBlocks,
                          (unsigned long long)plan->requiredWorkspace, plan->choice.estimateUs, (int)plan->fusedFold,
                          (unsigned long long)plan->gett.xcdTiles,
                          k == -2 ? "gett_wide_kernel" : k < 0 ? "gett_simple_kernel" : plan->choice.family == 2 ? "gett_gen_kernel" : plan->choice.family == 1 ? (tab[k].threads == 256 || tab[k].pf == 10 ? (tab[k].bk == 32 ? "gett_h16w4s_kernel" : tab[k].pf == 3 ? "gett_h16w4r_kernel" : tab[k].pf == 6 ? "gett_h16w4v_kernel" : tab[k].pf == 7 ? "gett_h16w4x_kernel" : tab[k].pf == 8 ? "gett_h16w4m_kernel" : tab[k].pf == 9 ? "gett_h16w4m4_kernel" : tab[k].pf == 10 ? "gett_h16w8m_kernel" : tab[k].pf == 11 ? "gett_h16w4q_kernel" : tab[k].pf == 12 ? "gett_h16w4p_kernel" : "gett_h16w4_kernel") : tab[k].pf == 4 ? "gett_h16s_kernel" : "gett_h16_kernel") : tab[k].fragPartials ? "gett_f32_stream_kernel" : "gett_f32_kernel");
        // contracted digits, fastest first: [extent, strideA, strideB]
        if (n > 0 && (size_t)n < len && k >= 0)     // 1: the kernel streams its operands with the nontemporal policy (no Infinity-Cache allocation)
            n += std::snprintf(buf + n, len - n, ",\"nt\":%d", tab[k].nt);
        if (n > 0 && (size_t)n < len && plan->choice.family == 2 && k >= 0)
            n += std::snprintf(buf + n, len - n, ",\"orientA\":%d,\"orientB\":%d,\"vec\":%d,\"elem\":%d", tab[k].layA, tab[k].layB, tab[k].vec, tab[k].elem);
        // 16-bit family: whether the launch is the RAG instantiation (masked / repaired last K-tile), and the strip plan — interior
        // [0, mInt) x [0, nInt) on `kernel`, the two edge strips as one launch of the 64 x 64 tile (ContractionChoice::stripKernel)
        if (n > 0 && (size_t)n < len && plan->choice.family == 1 && k >= 0)
            n += std::snprintf(buf + n, len - n, ",\"rag\":%d,\"strips\":%d,\"mInt\":%u,\"nInt\":%u",
                               (plan->gett.gK.total % 64u != 0u || (plan->gett.ragged & 1u)) ? 1 : 0, plan->choice.stripKernel >= 0 ? 1 : 0,
                               plan->choice.mInt, plan->choice.nInt);
        if (n > 0 && (size_t)n < len) n += std::snprintf(buf + n, len - n, ",\"Kdigits\":[");
        for (size_t i = 0; i < plan->view.K.size() && n > 0 && (size_t)n < len; ++i)
            n += std::snprintf(buf + n, len - n, "%s[%lld,%lld,%lld]", i ? "," : "", (long long)plan->view.K[i].extent,
                               (long long)plan->view.K[i].sA, (long long)plan->view.K[i].sB);
        if (n > 0 && (size_t)n < len) n += std::snprintf(buf + n, len - n, "]}");
    } else if (plan->kind == OpKind::Reduction && !plan->red.isPermutation) {
        n = std::snprintf(buf, len, "{\"op\":\"reduction\",\"variant\":%d,\"kept\":%u,\"red\":%u,\"splitR\":%u,\"workspace\":%llu}",
                          plan->red.variant, plan->red.p.kept.total, plan->red.p.red.total, plan->red.p.splitR,
                          (unsigned long long)plan->requiredWorkspace);
    } else {
        const EwPlan& e = (plan->kind == OpKind::Reduction) ? plan->red.perm : plan->ew;
        n = std::snprintf(buf, len, "{\"op\":\"elementwise\",\"variant\":%d,\"E0\":%u,\"E1\":%u,\"rest\":%u,\"blocks\":%u,\"tile0\":%u,\"order\":%u}",
                          e.variant, e.p.E0, e.p.E1, e.p.rest.total, e.p.nBlocks, e.p.tile0, e.p.order);
    }
    return n;
} CTAMD_API_CATCH_INT

// Diagnostics, all per handle.  Device buffer of 8 x uint64 per workgroup that the GETT kernel fills with phase
// timestamps (shader clock and wall clock); nullptr switches it off.
void ctamdSetTimingBuffer(cutensorHandle_t handle, void* deviceBuffer) try {
    if (handle != nullptr) handle->timingBuffer.store(static_cast<unsigned long long*>(deviceBuffer), std::memory_order_relaxed);
} CTAMD_API_CATCH_VOID

// enabled = 0 makes cutensorContract on this handle launch the GETT kernel only (the split-K partials stay unfolded, D is
// not written) so that a stream of back-to-back GETT launches can be timed with one event pair — per-launch event
// pairs put a ~6 us idle gap after every kernel and the chip leaves its steady clock state.  Never used by the samples.
void ctamdSetSplitKFold(cutensorHandle_t handle, int enabled) try {
    if (handle != nullptr) handle->skipFold.store(enabled == 0, std::memory_order_relaxed);
} CTAMD_API_CATCH_VOID

// Per-kernel timing of the GETT kernel inside cutensorContract on this handle: Begin() arms it, End() synchronises the
// recorded event pairs and returns the number of launches and their mean / min duration in ms.
void ctamdProfileBegin(cutensorHandle_t handle) try {
    if (handle == nullptr) return;
    std::lock_guard<std::mutex> g(handle->prof.mtx);
    handle->prof.events.clear();
    handle->prof.enabled.store(true, std::memory_order_relaxed);
} CTAMD_API_CATCH_VOID
int ctamdProfileEnd(cutensorHandle_t handle, float* meanMs, float* minMs) try {
    if (handle == nullptr) return 0;
    std::lock_guard<std::mutex> g(handle->prof.mtx);
    handle->prof.enabled.store(false, std::memory_order_relaxed);
    double sum = 0.0;
    float mn = 1e30f;
    int n = 0;
    for (auto& ev : handle->prof.events) {
        float t = 0.f;
        if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) {
            sum += t;
            mn = std::min(mn, t);
            ++n;
        }
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    handle->prof.events.clear();
    if (meanMs) *meanMs = n ? (float)(sum / n) : 0.f;
    if (minMs) *minMs = n ? mn : 0.f;
    return n;
} CTAMD_API_CATCH_INT

// Instantiated fp32 GETT kernels (test coverage bookkeeping): table size, and whether entry i is a
// measurement-only ablation variant (never planned unless CUTENSOR_AMD_ABLATION is set).
// 1 when the library was built with RESEARCH=1 (retired kernel families, TIMED / XST / EP instantiations, measurement switches)
int ctamdResearchKernelsBuilt(void) try {
#if defined(CTAMD_RESEARCH_KERNELS)
    return 1;
#else
    return 0;
#endif
} CTAMD_API_CATCH_INT

// 1 when this library reads the test / measurement switches (CTAMD_HOOK_ENV: the lib_hooks/ flavour and research builds)
int ctamdTestHooksBuilt(void) try { return CTAMD_HOOKS_BUILT; } CTAMD_API_CATCH_INT

int ctamdKernelCount(void) try {
    int count = 0;
    (void)gett_f32_kernels(&count);
    return count;
} CTAMD_API_CATCH_INT
int ctamdKernelIsAblation(int i) try {
    int count = 0;
    const GettKernelInfo* tab = gett_f32_kernels(&count);
    return (i >= 0 && i < count && tab[i].ablation) ? 1 : 0;
} CTAMD_API_CATCH_INT

// Number of ranked candidates for a contraction descriptor under a workspace limit (so that a
// caller can sweep CUTENSOR_PLAN_PREFERENCE_KERNEL_RANK / algo >= 0 exhaustively).
int ctamdCountCandidates(const cutensorHandle_t handle, const cutensorOperationDescriptor_t desc, uint64_t wsLimit) try {
    if (handle == nullptr || desc == nullptr || desc->kind != OpKind::Contraction) return -1;
    ContractionView v;
    if (build_contraction_view(*desc, v, nullptr) != CUTENSOR_STATUS_SUCCESS) return -1;
    if (v.dtype != HIP_R_32F || v.wide) return 0;
    return (int)rank_contraction_choices(v, wsLimit, handle->numCUs).size();
} CTAMD_API_CATCH_INT

}  // extern "C"
