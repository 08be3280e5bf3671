// plan_contraction.cpp — host planner for cutensorCreateContraction / cutensorCreatePlan.
//
// Turns "D_{modesD} = alpha * A_{modesA} * B_{modesB} + beta * C_{modesC}" (reference call site:
// cuTENSOR/contraction.cu:162-168, mode/extent conventions :46-59, packed column-major strides
// blocksparse.cu:80-81) into the GEMM view consumed by the gfx950 GETT kernels:
//
//   1. classify every mode label: L (in A, B and C), M (A and C), N (B and C), K (A and B);
//   2. drop extent-1 modes;
//   3. orient the problem so that the group holding D's stride-1 mode becomes kernel-N (the MFMA
//      output puts 16 consecutive lanes along n, so D stores are coalesced) — this may swap the
//      roles of A and B;
//   4. order the modes of each group by the stride of the tensor whose loads they drive, and fuse
//      neighbours that are jointly contiguous in every tensor that carries them ("mode fusion");
//   5. decide, per operand, whether 16-byte lanes can run along a free mode (LAY_F), along the
//      contracted mode (LAY_K), or not at all (LAY_S);
//   6. rank (tile shape, split-K) candidates with a small roofline cost model.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "internal.hpp"

namespace ctamd {

FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{};
    f.d = d;
    if (d < 2) {  // never used for division (extent-1 modes are dropped); keep it harmless
        f.magic = 0;
        f.shift = 0;
        return f;
    }
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;                     // s = ceil(log2 d) >= 1
    const uint64_t m = ((1ull << (31 + s)) / d) + 1;  // < 2^32, exact for n < 2^31
    f.magic = (uint32_t)m;
    f.shift = s - 1;                                  // q = mulhi(n, magic) >> (s-1)
    return f;
}

size_t dtype_size(hipDataType t) {
    switch (t) {
        case HIP_R_16F: case HIP_R_16BF: return 2;
        case HIP_R_32F: return 4;
        case HIP_R_64F: return 8;
        case HIP_C_32F: return 8;
        case HIP_C_64F: return 16;
        default: return 0;
    }
}

static int find_mode(const std::vector<int32_t>& modes, int32_t label) {
    for (size_t i = 0; i < modes.size(); ++i)
        if (modes[i] == label) return (int)i;
    return -1;
}

static bool has_duplicates(const std::vector<int32_t>& modes) {
    for (size_t i = 0; i < modes.size(); ++i)
        for (size_t j = i + 1; j < modes.size(); ++j)
            if (modes[i] == modes[j]) return true;
    return false;
}

// Fuse neighbours i, i+1 when every stride slot satisfies s[i+1] == s[i] * extent[i].
static void fuse_group(std::vector<CanonMode>& g, bool useA, bool useB, bool useCD) {
    std::vector<CanonMode> out;
    for (const CanonMode& m : g) {
        if (!out.empty()) {
            CanonMode& p = out.back();
            bool ok = true;
            if (useA) ok = ok && (m.sA == p.sA * p.extent);
            if (useB) ok = ok && (m.sB == p.sB * p.extent);
            if (useCD) ok = ok && (m.sC == p.sC * p.extent) && (m.sD == p.sD * p.extent);
            if (ok && p.extent * m.extent < (1ll << 31)) {
                p.extent *= m.extent;
                continue;
            }
        }
        out.push_back(m);
    }
    g.swap(out);
}

static uint64_t group_total(const std::vector<CanonMode>& g) {
    uint64_t t = 1;
    for (const CanonMode& m : g) t *= (uint64_t)m.extent;
    return t;
}

cutensorStatus_t build_contraction_view(const cutensorOperationDescriptor& op, ContractionView& v,
                                        std::string* why) {
    auto fail = [&](cutensorStatus_t st, const char* msg) {
        if (why) *why = msg;
        return st;
    };
    const TensorUse &A = op.A, &B = op.B, &C = op.C, &D = op.D;
    if (has_duplicates(A.modes) || has_duplicates(B.modes) || has_duplicates(C.modes))
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "repeated mode label inside one tensor");
    if (C.modes != D.modes) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "modes of C and D differ");
    if (C.desc.extent != D.desc.extent) return fail(CUTENSOR_STATUS_INVALID_VALUE, "extents of C and D differ");
    if (A.desc.dtype != B.desc.dtype || A.desc.dtype != C.desc.dtype || C.desc.dtype != D.desc.dtype)
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "mixed data types");
    if (A.op != CUTENSOR_OP_IDENTITY && A.op != CUTENSOR_OP_CONJ) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "opA");
    if (B.op != CUTENSOR_OP_IDENTITY && B.op != CUTENSOR_OP_CONJ) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "opB");
    if (C.op != CUTENSOR_OP_IDENTITY && C.op != CUTENSOR_OP_CONJ) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "opC");

    v = ContractionView{};
    v.dtype = A.desc.dtype;

    // gather distinct labels in a stable order: A's, then B's, then C's
    std::vector<int32_t> labels;
    auto add = [&](const std::vector<int32_t>& ms) {
        for (int32_t l : ms)
            if (std::find(labels.begin(), labels.end(), l) == labels.end()) labels.push_back(l);
    };
    add(A.modes); add(B.modes); add(C.modes);

    std::vector<CanonMode> gL, gFreeA, gFreeB, gK;   // free-of-A = in A and C, free-of-B = in B and C
    for (int32_t l : labels) {
        const int ia = find_mode(A.modes, l), ib = find_mode(B.modes, l), ic = find_mode(C.modes, l);
        CanonMode m;
        m.label = l;
        int64_t e = -1;
        auto take = [&](const TensorUse& T, int idx, int64_t& stride) -> bool {
            if (idx < 0) return true;
            if (e >= 0 && T.desc.extent[idx] != e) return false;
            e = T.desc.extent[idx];
            stride = T.desc.stride[idx];
            return true;
        };
        int64_t sa = 0, sb = 0, sc = 0, sd = 0;
        if (!take(A, ia, sa) || !take(B, ib, sb) || !take(C, ic, sc))
            return fail(CUTENSOR_STATUS_INVALID_VALUE, "a mode has different extents in different tensors");
        if (ic >= 0) sd = D.desc.stride[ic];
        m.extent = e; m.sA = sa; m.sB = sb; m.sC = sc; m.sD = sd;
        if (e == 1) continue;   // extent-1 modes carry no index
        if (e <= 0) return fail(CUTENSOR_STATUS_INVALID_VALUE, "non-positive extent");
        if (ia >= 0 && ib >= 0 && ic >= 0) gL.push_back(m);
        else if (ia >= 0 && ic >= 0) gFreeA.push_back(m);
        else if (ib >= 0 && ic >= 0) gFreeB.push_back(m);
        else if (ia >= 0 && ib >= 0) gK.push_back(m);
        else return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "mode appears in only one tensor");
    }

    // orientation: the free group that carries D's smallest stride becomes kernel-N
    int64_t bestA = INT64_MAX, bestB = INT64_MAX;
    for (const CanonMode& m : gFreeA) bestA = std::min(bestA, m.sD);
    for (const CanonMode& m : gFreeB) bestB = std::min(bestB, m.sD);
    v.swapped = (bestA < bestB);   // D's fastest free mode lives in A  => A plays kernel-B
    if (v.swapped) {
        for (auto* g : {&gL, &gFreeA, &gFreeB, &gK})
            for (CanonMode& m : *g) std::swap(m.sA, m.sB);
        v.M = gFreeB;   // kernel-A = user's B, its free modes
        v.N = gFreeA;
    } else {
        v.M = gFreeA;
        v.N = gFreeB;
    }
    v.K = gK;
    v.L = gL;

    auto bySA = [](const CanonMode& x, const CanonMode& y) { return x.sA < y.sA; };
    auto bySB = [](const CanonMode& x, const CanonMode& y) { return x.sB < y.sB; };
    auto bySD = [](const CanonMode& x, const CanonMode& y) { return x.sD < y.sD; };
    std::stable_sort(v.M.begin(), v.M.end(), bySA);
    std::stable_sort(v.N.begin(), v.N.end(), bySD);
    std::stable_sort(v.L.begin(), v.L.end(), bySD);
    // K: follow whichever operand is K-contiguous (A first), else A's order.
    bool kContigA = false, kContigB = false;
    for (const CanonMode& m : v.K) { kContigA |= (m.sA == 1); kContigB |= (m.sB == 1); }
    // CUTENSOR_AMD_KORDER (experiment knob): "A" / "B" = follow that operand's strides; otherwise an
    // explicit digit list, fastest first, e.g. "d,c:4,b,c" = mode d, the inner 4 of mode c, mode b, the
    // rest of c (labels as characters).  Any order of the contracted digits is a valid GETT view; the
    // order decides how a K slice maps to memory in A and B.
    const char* korder = ctamd_research_env("CUTENSOR_AMD_KORDER");
    bool followB = (!kContigA && kContigB);
    bool custom = false;
    if (korder && korder[0] == 'B' && korder[1] == 0) followB = true;
    else if (korder && korder[0] == 'A' && korder[1] == 0) followB = false;
    else if (korder && korder[0]) {
        std::vector<CanonMode> rest = v.K, out;
        bool ok = true;
        for (const char* c = korder; *c && ok;) {
            const int32_t label = (int32_t)(unsigned char)*c++;
            int64_t inner = 0;
            if (*c == ':') { ++c; inner = std::strtoll(c, const_cast<char**>(&c), 10); }
            if (*c == ',') ++c;
            auto it = std::find_if(rest.begin(), rest.end(), [&](const CanonMode& m) { return m.label == label; });
            if (it == rest.end()) { ok = false; break; }
            if (inner > 1 && inner < it->extent && it->extent % inner == 0) {
                CanonMode lo = *it;
                lo.extent = inner;
                out.push_back(lo);
                it->extent /= inner; it->sA *= inner; it->sB *= inner;
            } else {
                out.push_back(*it);
                rest.erase(it);
            }
        }
        if (ok && rest.empty()) { v.K = out; custom = true; }
    }
    if (custom) { /* keep the requested order */ }
    else if (followB) std::stable_sort(v.K.begin(), v.K.end(), bySB);
    else         std::stable_sort(v.K.begin(), v.K.end(), bySA);

    fuse_group(v.M, true, false, true);
    fuse_group(v.N, false, true, true);
    fuse_group(v.K, true, true, false);
    fuse_group(v.L, true, true, true);

    v.totL = group_total(v.L); v.totM = group_total(v.M);
    v.totN = group_total(v.N); v.totK = group_total(v.K);
    const uint64_t lim = (1ull << 31) - 1;
    if (v.totK > lim) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "the contracted modes span more than 2^31-1 elements");
    if (v.totL > lim || v.totM > lim || v.totN > lim || (int)v.L.size() > kMaxGroupModes || (int)v.M.size() > kMaxGroupModes ||
        (int)v.N.size() > kMaxGroupModes || (int)v.K.size() > kMaxGroupModes) {
        // more digits than the tiled kernels' argument block describes: the mode-table kernel takes it
        if (v.L.size() + v.M.size() + v.N.size() + v.K.size() > 128) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "more than 128 unfusable modes");
        v.wide = true;
        v.layA = v.layB = LAY_S;
        return CUTENSOR_STATUS_SUCCESS;
    }
    v.alignA = v.swapped ? op.B.desc.alignment : op.A.desc.alignment;
    v.alignB = v.swapped ? op.A.desc.alignment : op.B.desc.alignment;
    v.alignD = op.D.desc.alignment;
    if (v.dtype == HIP_C_32F || v.dtype == HIP_C_64F) {   // complex data: the general MFMA family (pick_gen_choice) decides its own lanes
        v.layA = v.layB = LAY_S;
        return CUTENSOR_STATUS_SUCCESS;
    }

    // ---- operand layouts: 16-byte lanes = 4 fp32 or 8 bf16/fp16 elements ------------------------
    const int64_t vec = (dtype_size(v.dtype) == 2) ? 8 : 4;
    auto all_mult4_except = [vec](const std::vector<const std::vector<CanonMode>*>& groups, bool slotA,
                                  const CanonMode* except) {
        for (auto* g : groups)
            for (const CanonMode& m : *g) {
                if (&m == except) continue;
                const int64_t s = slotA ? m.sA : m.sB;
                if (s % vec != 0) return false;
            }
        return true;
    };
    const uint32_t alignA = v.swapped ? op.B.desc.alignment : op.A.desc.alignment;
    const uint32_t alignB = v.swapped ? op.A.desc.alignment : op.B.desc.alignment;
    // 16-bit data (round 6): the LDS-DMA kernels take 16-byte units at ANY 2-byte address (buffer_load_dwordx4 ... lds lands the right
    // bytes whatever the alignment: tools/ubench/ldsdma_unaligned.hip, profiles/r06a_*; 29 B/clk/CU at 4 or 8 (mod 16), 25.6 at 2 (mod 16),
    // 31 aligned), so an operand qualifies by its stride-1 mode alone.  An extent of that mode that is not a multiple of 8 leaves a partial
    // last unit: the kernels' RAG instantiations handle it (x_rag_mask / x_rag_fix) when the unit cannot straddle a digit boundary, i.e.
    // when the mode is alone in its group.
    auto pick16 = [&](bool slotA, const std::vector<CanonMode>& freeG) {
        const std::vector<CanonMode>* groups[3] = {&freeG, &v.K, &v.L};
        for (const std::vector<CanonMode>* g : groups)
            for (const CanonMode& m : *g)
                if ((slotA ? m.sA : m.sB) < 0) return (int)LAY_S;
        if (!v.K.empty()) {
            const CanonMode& k0 = v.K.front();
            if ((slotA ? k0.sA : k0.sB) == 1 && (k0.extent % 8 == 0 || v.K.size() == 1)) return (int)LAY_K;
        }
        if (!freeG.empty()) {
            const CanonMode& f0 = freeG.front();
            if ((slotA ? f0.sA : f0.sB) == 1 && (f0.extent % 8 == 0 || freeG.size() == 1)) return (int)LAY_F;
        }
        return (int)LAY_S;
    };
    if (dtype_size(v.dtype) == 2) {
        v.layA = pick16(true, v.M);
        v.layB = pick16(false, v.N);
        return CUTENSOR_STATUS_SUCCESS;
    }
    auto pick = [&](bool slotA, const std::vector<CanonMode>& freeG, uint32_t align) {
        if (align % 16 != 0) return (int)LAY_S;
        std::vector<const std::vector<CanonMode>*> groups = {&freeG, &v.K, &v.L};
        if (!v.K.empty()) {
            const CanonMode& k0 = v.K.front();
            const int64_t s = slotA ? k0.sA : k0.sB;
            if (s == 1 && k0.extent % vec == 0 && all_mult4_except(groups, slotA, &k0)) return (int)LAY_K;
        }
        if (!freeG.empty()) {
            const CanonMode& f0 = freeG.front();
            const int64_t s = slotA ? f0.sA : f0.sB;
            if (s == 1 && f0.extent % vec == 0 && all_mult4_except(groups, slotA, &f0)) return (int)LAY_F;
        }
        return (int)LAY_S;
    };
    v.layA = pick(true, v.M, alignA);
    v.layB = pick(false, v.N, alignB);
    v.lanesA = v.layA != LAY_S;
    v.lanesB = v.layB != LAY_S;
    // fp32 (round 6): like 16-bit data, the LDS-DMA ring kernels (gett_f32_stream.hip) stage 16-byte units from any 4-byte address, so
    // the stride-1 mode alone decides the layout; a ragged extent of that mode needs it to be alone in its group.  The register-staged
    // kernels keep the strict rule (lanesA / lanesB).
    auto pick32 = [&](bool slotA, const std::vector<CanonMode>& freeG) {
        const std::vector<CanonMode>* groups[3] = {&freeG, &v.K, &v.L};
        for (const std::vector<CanonMode>* g : groups)
            for (const CanonMode& m : *g)
                if ((slotA ? m.sA : m.sB) < 0) return (int)LAY_S;
        if (!v.K.empty()) {
            const CanonMode& k0 = v.K.front();
            if ((slotA ? k0.sA : k0.sB) == 1 && (k0.extent % 4 == 0 || v.K.size() == 1)) return (int)LAY_K;
        }
        if (!freeG.empty()) {
            const CanonMode& f0 = freeG.front();
            if ((slotA ? f0.sA : f0.sB) == 1 && (f0.extent % 4 == 0 || freeG.size() == 1)) return (int)LAY_F;
        }
        return (int)LAY_S;
    };
    if (v.dtype == HIP_R_32F) {
        if (!v.lanesA) v.layA = pick32(true, v.M);
        if (!v.lanesB) v.layB = pick32(false, v.N);
    }
    return CUTENSOR_STATUS_SUCCESS;
}

// fp32 ring kernels: the launch needs the RAG instantiation (gett_f32_stream.hip) — ragged K, or an operand without strict 16-byte lanes
// (a partial unit may reach past the end of the tensor: the RAG descriptors end with it)
static bool f32_needs_rag(const ContractionView& v) {
    return v.totK % 32 != 0 || !v.lanesA || !v.lanesB;
}

// ---------------------------------------------------------------------------------------------
// Cost model.  Constants are gfx950 figures from the microarchitecture guide; the model only has
// to order candidates, not predict wall time.
// ---------------------------------------------------------------------------------------------
static double tile_efficiency(const GettKernelInfo& k) {
    // fraction of MFMA issue a resident workgroup of this shape sustains (measured on MI355X,
    // see profiles/): small tiles read more LDS bytes per flop and expose more barrier time.
    const int area = k.bm * k.bn;
    // streaming kernels: LDS-DMA ring, prefetched fragments; the 3-deep ring measures ~2 % faster than the 4- and
    // 6-deep ones in the device's steady clock state (headline einsum 42.5 vs 43.6 us per step)
    // (compute-bound problems: the 4-deep ring — 4096^3 128 x 128: 148 TFLOP/s against 130 on the 3-deep one, 4098^3 alike,
    // profiles/r06i_f32_candidates_4098_4096.jsonl; memory-bound ones tie here and the 3-deep ring wins the tie-break below)
    if (k.fragPartials) return area >= 96 * 96 ? (k.pf == 4 ? 0.93 : 0.92) : 0.75;
    if (area >= 128 * 128) return 0.85;
    if (area >= 96 * 96) return 0.80;
    if (area >= 64 * 64) return 0.65;
    if (area >= 48 * 48) return 0.70;
    return 0.45;
}

std::vector<ContractionChoice> rank_contraction_choices(const ContractionView& v, uint64_t wsLimit,
                                                        int numCUs, bool operandsStreamed) {
    std::vector<ContractionChoice> out;
    int count = 0;
    const GettKernelInfo* tab = gett_f32_kernels(&count);
    const double clk = 2.4e9, flopPerClkCU = 256.0;
    const double hbm = 6.0e12, l2bw = 20.0e12;
    const double M = (double)v.totM, N = (double)v.totN, K = (double)v.totK, L = (double)v.totL;

    const bool withAblations = ctamd_research_env("CUTENSOR_AMD_ABLATION") != nullptr;
    // byte span of an operand beyond its batch offset: the streaming kernels address it through a buffer
    // descriptor with 32-bit byte offsets
    auto span_bytes = [&](bool slotA) {
        uint64_t n = 1;
        for (const std::vector<CanonMode>* g : {slotA ? &v.M : &v.N, &v.K})
            for (const CanonMode& m : *g) n += (uint64_t)(m.extent - 1) * (uint64_t)std::llabs(slotA ? m.sA : m.sB);
        return n * 4ull;
    };
    const bool fits32 = span_bytes(true) < (1ull << 32) - (1ull << 20) && span_bytes(false) < (1ull << 32) - (1ull << 20);
    // with the batch modes: what a RAG descriptor (base = operand + batch offset) and a masked lane (bit 31) must stay below
    auto full_span_bytes = [&](bool slotA) {
        uint64_t n = span_bytes(slotA) / 4ull;
        for (const CanonMode& m : v.L) n += (uint64_t)(m.extent - 1) * (uint64_t)std::llabs(slotA ? m.sA : m.sB);
        return n * 4ull;
    };
    const bool fits31 = full_span_bytes(true) < (1ull << 31) - (1ull << 20) && full_span_bytes(false) < (1ull << 31) - (1ull << 20);
    const bool needRag = f32_needs_rag(v);
    for (int i = 0; i < count; ++i) {
        const GettKernelInfo& k = tab[i];
        if (k.ablation && !withAblations) continue;
        if (k.fragPartials && !fits32) continue;
        // a kernel is usable if each operand admits its layout (LAY_S kernels take anything).  The register-staged kernels see an
        // operand without strict 16-byte lanes as LAY_S; the ring kernels (fragPartials) take the relaxed layouts — their RAG twin runs
        // when K is ragged (one contracted mode) or an operand has no strict lanes: spans below 2^31 bytes, no nontemporal form
        const bool rag = k.fragPartials && needRag;
        const int effA = (k.fragPartials || v.lanesA) ? v.layA : (int)LAY_S, effB = (k.fragPartials || v.lanesB) ? v.layB : (int)LAY_S;
        const bool okA = (k.layA == effA) || (k.layA == LAY_S);
        const bool okB = (k.layB == effB) || (k.layB == LAY_S);
        if (!okA || !okB) continue;
        if ((k.layA == LAY_S) != (k.layB == LAY_S)) continue;   // table only holds S/S pairs
        if (k.layA == LAY_S && v.lanesA && v.lanesB) continue;  // vector kernels exist
        if (rag && (!fits31 || k.nt || k.ablation || (v.K.size() > 1 && v.totK % k.bk != 0))) continue;
        if (k.kfast && !(rag && v.K.size() == 1) && (v.K.empty() || (v.K.front().extent % k.bk) != 0)) continue;   // tile would straddle a K-mode period

        const uint64_t tilesM = (v.totM + k.bm - 1) / k.bm, tilesN = (v.totN + k.bn - 1) / k.bn;
        const uint64_t tiles = tilesM * tilesN * v.totL;
        // nontemporal operand stream: only when every operand byte is read exactly once (one tile row and one tile column) AND the
        // operands are well beyond the 256-MiB Infinity Cache (> 1.4 x), i.e. back-to-back calls cannot find them on-die anyway (measured:
        // headline shape 201 MB: +3.6 % from HBM, -5.5 % cache-resident; b = 96, 302 MB: -5 %; b = 128, 403 MB: +6 %; profiles/r03_headline_nt.txt)
        static const bool ntAnySize = CTAMD_HOOK_ENV("CUTENSOR_AMD_NT") != nullptr;   // tests: exercise the nt kernels on small read-once shapes
        // ... or the caller says so: CUTENSOR_AMD_PLAN_PREFERENCE_OPERANDS_STREAMED (include/cutensor/types.h) — "every call finds its operands in
        // HBM", which the library cannot know for the 201 MB of the headline einsum
        if (k.nt && !(tilesM == 1 && tilesN == 1 && (ntAnySize || operandsStreamed || 4.0 * L * (M * K + N * K) > 1.4 * 256.0 * 1024.0 * 1024.0))) continue;
        const uint64_t kTiles = (v.totK + k.bk - 1) / k.bk;
        // split-K candidates: 1, and powers of two up to what keeps >= 4 K-tiles per slice
        std::vector<uint32_t> splits = {1};
        for (uint32_t s = 2; s <= 1024; s *= 2) {
            if (kTiles / s < 4) break;
            if (tiles * s > (uint64_t)numCUs * 16) break;
            splits.push_back(s);
        }
        for (uint32_t s : splits) {
            ContractionChoice c;
            c.kernel = i;
            if (k.fragPartials && v.totK % k.bk != 0 && !rag) continue;   // whole K-tiles (or the RAG twin's masked last one); any number of tiles per slice
            const uint64_t tilesPerSlice = (kTiles + s - 1) / s;
            c.kPerSlice = (uint32_t)(tilesPerSlice * k.bk);
            c.splitK = (uint32_t)((v.totK + c.kPerSlice - 1) / c.kPerSlice);
            if (c.splitK < 1) c.splitK = 1;
            if (c.splitK != s && s != 1) continue;   // rounding collapsed this candidate
            if (k.fragPartials)   // accumulator-order partials cover whole (padded) tiles
                c.workspace = (c.splitK > 1) ? (uint64_t)c.splitK * tiles * k.bm * k.bn * 4ull : 0ull;
            else
                c.workspace = (c.splitK > 1) ? (uint64_t)c.splitK * v.totL * v.totM * v.totN * 4ull : 0ull;
            if (c.workspace > wsLimit) continue;

            const double blocks = (double)tiles * c.splitK;
            const double wgPerCU = k.fragPartials ? 1.0 : 2.0;   // streaming kernels own the CU's LDS
            const double slots = numCUs * wgPerCU;
            const double rounds = std::ceil(blocks / slots);
            const double flopsBlock = 2.0 * k.bm * k.bn * (double)c.kPerSlice;
            // a CU shares its MFMA pipes between resident workgroups
            const double tBlock = flopsBlock / (flopPerClkCU * clk * tile_efficiency(k) / wgPerCU);
            const double occupancyFill = std::min(1.0, blocks / slots);
            double tCompute = rounds * tBlock * (occupancyFill < 1.0 && blocks < numCUs ? 0.5 : 1.0);
            if (blocks <= numCUs) tCompute = flopsBlock / (flopPerClkCU * clk * tile_efficiency(k));
            // traffic: every tile row re-reads its A panel, every tile column its B panel
            const double bytesA = 4.0 * L * (double)tilesM * k.bm * K * (double)tilesN;
            const double bytesB = 4.0 * L * (double)tilesN * k.bn * K * (double)tilesM;
            const double bytesUnique = 4.0 * L * (M * K + N * K + M * N);
            const double bytesPartial = (c.splitK > 1) ? 2.0 * (double)c.workspace : 0.0;
            const double tMem = std::max((bytesA + bytesB) / l2bw, (bytesUnique + bytesPartial) / hbm);
            // the fold of accumulator-order partials (splitk_reduce_frag_kernel) gives 32 slice groups to every output quad: with few slices
            // most of its lanes idle — 4098^3 in two slices folds 134 MB in ~400 us (profiles/r06i_f32_candidates_4098_4096.jsonl), the
            // headline's 256 slices fold 9.4 MB in 4.7 us
            // (only the EXTRA time of the idle lanes is charged: at 32 slices and beyond the term vanishes and the ranking of round 5 stands)
            const double tFold = (c.splitK > 1 && c.splitK < 32 && k.fragPartials) ? (double)c.workspace / 5.0e12 * (32.0 / c.splitK - 1.0) : 0.0;
            // What a workgroup costs outside its K-tiles (setup, first loads, epilogue), per ROUND of workgroups (round 6): measured on short
            // contracted ranges, where it decides — 16384^2 x 128 on 128 x 128 tiles: 64 rounds in 967 us with 7.4 us of MFMA work each;
            // 8192 x 128 x 8192: the 32-slice candidate (8 rounds) 216 us where this model said 122 and the unsplit 64 x 64 one (one round)
            // 121 (profiles/r06zk_tune_*.jsonl).  ~2 us + ~0.34 us per 1024 tile elements; a single round pays it once like everybody else.
            // Ring kernels whose LDS is at most half a CU's (64 x 64 on four K-tiles, 96 x 96 on three: 64 / 72 KiB) run two workgroups per
            // CU and overlap one's fixed cost with the other's K-tiles (16384^2 x 128: 96 x 96 ring 3 836 us, 128 x 128 ring 4 967).  Long
            // contracted ranges have these costs inside tile_efficiency() already (calibrated at >= 64 K-tiles): only the excess is charged.
            const double ldsBytes = k.fragPartials ? (double)k.pf * (k.bm + k.bn) * k.bk * 4.0 : 0.0;
            const double overlap = (!k.fragPartials || ldsBytes <= 80.0 * 1024.0) ? 2.0 : 1.0;
            const double roundsEff = std::ceil(blocks / (numCUs * overlap));
            const double shortK = std::max(0.0, 1.0 - (double)tilesPerSlice / 64.0);
            const double tRound = (roundsEff > 1.0) ? (roundsEff - 1.0) * (2.0e-6 + 0.34e-6 * (double)(k.bm * k.bn) / 1024.0) * shortK : 0.0;
            const double tFix = ((c.splitK > 1) ? 3.0e-6 + tFold : 0.0) + tRound;
            c.estimateUs = (std::max(tCompute, tMem) + tFix + 2.0e-6) * 1e6;
            if (k.fragPartials && k.pf == 3) c.estimateUs *= 0.999;   // tie-break for memory-bound estimates: the 3-deep ring wins by ~2 %
            if (k.nt) c.estimateUs *= 0.96;                            // eligible (see above): ahead of its default-policy twin
            out.push_back(c);
        }
    }
    std::stable_sort(out.begin(), out.end(),
                     [](const ContractionChoice& a, const ContractionChoice& b) { return a.estimateUs < b.estimateUs; });
    return out;
}

// 16-bit data (bf16 / fp16, fp32 accumulation) on the aligned LDS-DMA family (gett_h16v.hip, gett_h16p.hip: 256 x 256 one-tile and
// persistent, 128 x 128 on a ring of two / four K-tiles, 64 x 64; K-tile 64): picked when both operands admit 16-byte lanes, one tile of
// an operand spans less than 2 GiB, and the fastest contracted mode holds whole 64-deep K-tiles — or is the ONLY contracted mode
// (ragged K: the masked last K-tile of the RAG instantiations).  Returns false -> the general MFMA family (pick_gen_choice).
// The launch needs the kernels' RAG instantiation (masked last K-tile of the last slice): a ragged contracted range, or a free-contiguous
// operand whose stride-1 mode is not a multiple of 8 long (its last row-unit can reach past the end of the tensor: x_rag_mask).
// Sweep-ragged K (round 6): SEVERAL contracted modes and the fastest one does not hold whole 64-deep K-tiles ('abcd,dcbe->ae' at extents
// of 96 or 40).  The RAG instantiations of the 256 x 256, 128 x 128 and 64 x 64 kernels count K-tiles in the padded space — ceil(E0 / 64) tiles per
// sweep of the fastest mode — and stage the last tile of EVERY sweep with the lanes past the end of the mode out of range (x_rag_toggle:
// the mask is switched on and off as the odometer goes).  Admitted when no 16-byte unit is partial or can reach past the tensor: a
// K-contiguous operand has E0 % 8 == 0 by its layout class (pick16), a free-contiguous one needs a stride-1 extent that is a multiple of 8.
// (Partial units — d = 50 — were built too: the tail of the unit zeroed in LDS behind an extra workgroup barrier in every sweep's last
// tile, `past` units patched in the very last one.  Parity green, but the per-tile repair state in the main loop cost EVERY launch of the
// RAG instantiations 10-20 % — 4096^2 x 4104 96-110 -> 111-133 us — and 'abcd,dcbe->ae' with d = 50 ran at 146 TFLOP/s against the
// general family's 233: profiles/r06zz8_*; removed.)
static bool h16_sweep_ragged(const ContractionView& v) {
    if (v.K.size() < 2 || v.K.front().extent % 64 == 0) return false;
    if (v.layA == LAY_K || v.layB == LAY_K) { if (v.K.front().extent % 8 != 0) return false; }
    if (v.layA == LAY_F && (v.M.empty() || v.M.front().extent % 8 != 0)) return false;
    if (v.layB == LAY_F && (v.N.empty() || v.N.front().extent % 8 != 0)) return false;
    return true;
}
// share of a sweep's K-tiles that holds data
static double h16_sweep_fill(const ContractionView& v) {
    const int64_t e0 = v.K.front().extent;
    return (double)e0 / (64.0 * (double)((e0 + 63) / 64));
}
// K-tiles of the 16-bit LDS-DMA family (padded per sweep of the fastest contracted mode when that one is ragged)
static uint64_t h16_k_tiles(const ContractionView& v) {
    if (h16_sweep_ragged(v)) return (uint64_t)((v.K.front().extent + 63) / 64) * (v.totK / (uint64_t)v.K.front().extent);
    return (v.totK + 63) / 64;
}

static bool h16_needs_rag(const ContractionView& v) {
    if (v.totK % 64 != 0 || h16_sweep_ragged(v)) return true;
    if (v.layA == LAY_F && !v.M.empty() && v.M.front().extent % 8 != 0) return true;
    if (v.layB == LAY_F && !v.N.empty() && v.N.front().extent % 8 != 0) return true;
    return false;
}

bool pick_h16_choice(const ContractionView& v, uint64_t wsLimit, int numCUs, ContractionChoice& c) {
    if (v.dtype != HIP_R_16BF && v.dtype != HIP_R_16F) return false;
    if ((v.layA != LAY_K && v.layA != LAY_F) || (v.layB != LAY_K && v.layB != LAY_F)) return false;
    if (v.K.empty()) return false;
    // Whole 64-deep K-tiles in the fastest contracted mode — or (round 5) ONE contracted mode of any extent the 16-byte lanes admit
    // (a K-contiguous operand has extent % 8 == 0 by its layout class, a free-contiguous one takes any extent): the four-wave kernel
    // stages the last K-tile with the lanes past the end of the mode out of range (gett_h16w4x_kernel<..., RAG = true>).
    // ... or (round 6) several contracted modes with a ragged fastest one: the masked last K-tile of every sweep (h16_sweep_ragged).
    const bool sweep = h16_sweep_ragged(v);
    if (!sweep && ((v.totK % 64 != 0) ? (v.K.size() != 1) : (v.K.front().extent % 64 != 0))) return false;
    if (sweep && h16_k_tiles(v) >= (1ull << 30)) return false;     // (GettParams::ragged carries the padded tile count in 30 bits)
    // sweeps that fill less than 45 % of their K-tiles ('abcd,dcbe->ae' with d = 16: a quarter) are no faster than the general family on its
    // 16-byte lanes (measured, profiles/r06zz6_*: d = 40, 62 %: 543 against 411 TFLOP/s; d = 16, 25 %: below its 360) — cutensorCreatePlan
    // then looks at copying an operand so that the contracted modes fuse (api.cpp plan_repack), else the general family takes it
    if (sweep && h16_sweep_fill(v) < 0.45 && !CTAMD_HOOK_ENV("CUTENSOR_AMD_H16_WAVES")) return false;
    const bool ragged = h16_needs_rag(v);         // the candidates are the kernels that have a RAG instantiation
    // The 16-bit kernels address an operand with 32-bit byte offsets relative to a 64-bit base that moves with the workgroup
    // tile, the wave and the K-tile (gett_h16.hip, HOperand / HOdometer): what has to stay below 2^31 bytes is the span of
    // ONE 256-row x 64-k tile, whatever the size of the tensor.
    auto tile_span_bytes = [&](bool slotA) {
        uint64_t n = 1;
        for (const CanonMode& m : (slotA ? v.M : v.N))
            n += (uint64_t)(std::min<int64_t>(m.extent, 256) - 1) * (uint64_t)std::llabs(slotA ? m.sA : m.sB);
        n += 64ull * (uint64_t)std::llabs(slotA ? v.K.front().sA : v.K.front().sB);
        return n * 2ull;
    };
    if (tile_span_bytes(true) >= (1ull << 31) || tile_span_bytes(false) >= (1ull << 31)) return false;
    for (const std::vector<CanonMode>* g : {&v.M, &v.N, &v.K})       // relative offsets are unsigned
        for (const CanonMode& m : *g)
            if (m.sA < 0 || m.sB < 0) return false;
    int count = 0;
    const GettKernelInfo* tab = gett_h16_kernels(&count);
    c = ContractionChoice{};
    c.family = 1;
    c.kernel = (v.dtype == HIP_R_16BF ? 0 : 4) + (v.layA == LAY_F ? 2 : 0) + (v.layB == LAY_F ? 1 : 0);
    // entries 0..7: eight waves, two rows alternated by barriers (ping-pong); 8..15: four waves per workgroup (one per
    // SIMD); 16..23: eight free-running waves, K-tile of 32, deep LDS ring, one barrier per K-tile; 24..31: four waves on that
    // ring; 32..39: four waves, register-staged; 40..47: four waves, lean instruction stream (gett_h16v.hip); 48..55: the same on
    // the 16x16x32 MFMA — the default since round 3 (+8-14 % under the power limit on every layout)
    static const int variant = [] {
        const char* e = CTAMD_HOOK_ENV("CUTENSOR_AMD_H16_WAVES");
        if (e && e[0] == '4' && e[1] == 's') return 24;
        if (e && e[0] == '4' && e[1] == 'r') return 32;
        if (e && e[0] == '4' && e[1] == 'v') return 40;
        if (e && e[0] == '4' && e[1] == 'x') return 48;
        if (e && e[0] == '4' && e[1] == 'p') return 88;
        if (e && e[0] == '4' && e[1] == 'q') return 80;
        if (e && e[0] == '8' && e[1] == 'm') return 72;
        if (e && e[0] == '4' && e[1] == 'm' && e[2] == '4') return 64;
        if (e && e[0] == '4' && e[1] == 'm') return 56;
        if (e && e[0] == '4') return 8;
        if (e && e[0] == 's') return 16;
        if (e && (e[0] == '8' || e[0] == 'p')) return 0;
        return 48;
    }();
    const int layoutIdx = c.kernel;
    const uint64_t kTiles = h16_k_tiles(v);
    const uint64_t perSliceBytes = v.totL * v.totM * v.totN * 4ull;
    const bool forced = CTAMD_HOOK_ENV("CUTENSOR_AMD_H16_WAVES") != nullptr;
    auto tiles_of = [&](int var) {
        const GettKernelInfo& k = tab[layoutIdx + var];
        return std::ceil((double)v.totM / k.bm) * std::ceil((double)v.totN / k.bn) * (double)v.totL;
    };
    // workgroup slots of the chip: the 128 x 128 kernel on the two-deep ring (variant 56) runs two workgroups per CU
    auto slots_of = [&](int var) { return (double)numCUs * ((var == 56 || var == 80) ? 2.0 : 1.0); };
    // split-K when the output tiles alone leave most CUs idle: slices of >= 4 K-tiles, fp32 partials [slice][L][M][N]
    auto auto_split = [&](int var) -> uint64_t {
        const double tiles = tiles_of(var), slots = slots_of(var);
        uint64_t split = 1;
        if (tiles * 2.0 <= slots && kTiles >= 8) {
            split = std::min<uint64_t>((uint64_t)(slots / tiles), kTiles / 4);
            while (split > 1 && split * perSliceBytes > wsLimit) --split;
            if (split < 2) split = 1;
        }
        return split;
    };
    // Time model (us) of a variant at a split, from the shape sweeps of round 4 (profiles/r04*_h16_shape_sweep*: 8192^3 .. 256^2 x 16384 on
    // all variants): rounds x (fixed cost per workgroup + K-tiles x time per K-tile) + the fold.  Only has to ORDER the candidates.
    //   256 x 256, four waves (48):  10 us fixed, 1.00 us per K-tile      (256 x 256, eight waves (0): 10 us fixed, 1.13 us per K-tile)
    //   128 x 128, ring 2 (56), two workgroups per CU: 5 us fixed, 0.92 us per K-tile (0.68 with the CU to itself)
    //   128 x 128, ring 4 (64), one workgroup per CU:  4.5 us fixed, 0.46 us per K-tile
    //   64 x 64 (80), two workgroups per CU: 4.6 us fixed, 0.31 us per K-tile (4.4 / 0.223 while every workgroup has a CU to itself;
    //   profiles/r04w_sweep_4q_*: 1024^3 7.6-7.9 us, 4096^3 195-202 us)
    //   fold (splitk_reduce_wide_kernel): 3 us + partial bytes written and read back at ~5 TB/s
    auto model_tiles_us = [&](int var, double tiles, uint64_t split) {
        const double wgs = tiles * (double)split, slots = slots_of(var);
        const double kt = std::ceil((double)kTiles / (double)split);
        double fix, per;
        if (var == 64) { fix = 4.5; per = 0.46; }
        else if (var == 80) { fix = wgs <= (double)numCUs ? 4.4 : 4.6; per = wgs <= (double)numCUs ? 0.223 : 0.31; }
        else if (var == 56) { fix = 5.0; per = wgs <= (double)numCUs ? 0.68 : 0.92; }
        else { fix = 10.0; per = 1.0; }
        double t = std::ceil(wgs / slots) * (fix + kt * per);
        // the persistent form of the 256 x 256 kernel (88, gett_h16p.hip, round 5): a workgroup walks its tiles, interior tiles with an even
        // K-tile count stream into each other (the next tile's first K-tiles are fetched by the last K-tile bodies) — the first tile costs
        // what gett_h16w4x_kernel's does plus ~0.5 us of extra setup, every further one ~6 us less (profiles/r05e_h16p_vs_4x.jsonl: 8192^2 x
        // 512 / 1024 / 2048 / 4096 / 8192 +6.8 / +3.3 / +2.5 / +1.8 / +0.8 %, one-round shapes -1 %)
        if (var == 88) { const double ke = kt + (std::fmod(kt, 2.0) != 0.0 ? 1.0 : 0.0); t = 10.5 + ke * per + (std::ceil(wgs / slots) - 1.0) * (4.5 + ke * per); }
        if (split > 1) t += 3.0 + 2.0 * (double)split * (double)perSliceBytes / 5.0e6;
        return t;
    };
    auto model_us = [&](int var, uint64_t split) { return model_tiles_us(var, tiles_of(var), split); };
    // Strip plan of a candidate (round 6): its whole tiles as the interior launch, the two edge strips as ONE launch of the 64 x 64 kernel
    // (entry 80: two tile rectangles in one grid, GettParams::tilesM2).  Worth it when the partial tiles push the launch into another
    // round: 4100^3 on 256 x 256 tiles = 289 tiles = two rounds (150 us by this model) against 256 tiles + 129 strip tiles (~97 us).
    // One M and one N mode only (a strip is a contiguous index range of a mode group, but the interior's whole-tile test and the
    // kernels' edge clamps are written for it), no split-K.  Returns the model time, 1e30 when the candidate has no strip form.
    auto strip_us = [&](int cand, uint32_t& mInt, uint32_t& nInt) {
        if (layoutIdx + 80 >= count || layoutIdx + cand >= count) return 1e30;
        const GettKernelInfo& k = tab[layoutIdx + cand];
        if (k.bm <= 64) return 1e30;
        mInt = (uint32_t)(v.totM / k.bm) * (uint32_t)k.bm;
        nInt = (uint32_t)(v.totN / k.bn) * (uint32_t)k.bn;
        if ((mInt == v.totM && nInt == v.totN) || mInt == 0 || nInt == 0) return 1e30;
        const double tilesInt = (double)(mInt / k.bm) * (double)(nInt / k.bn) * (double)v.totL;
        const double strip = (std::ceil((double)(v.totM - mInt) / 64.0) * std::ceil((double)v.totN / 64.0) +
                              std::ceil((double)mInt / 64.0) * std::ceil((double)(v.totN - nInt) / 64.0)) * (double)v.totL;
        return model_tiles_us(cand, tilesInt, 1) + model_tiles_us(80, strip, 1) + 1.5;   // + the gap between the two launches
    };
    static const bool noStrips = [] { const char* e = CTAMD_HOOK_ENV("CUTENSOR_AMD_H16_STRIPS"); return e && e[0] == '0'; }();
    const bool stripsOK = !noStrips && v.M.size() == 1 && v.N.size() == 1;
    int var = variant;
    uint64_t split = 1;
    const bool usable = forced && layoutIdx + var < count && tab[layoutIdx + var].ablation != 2;   // a retired family asked for in a production build: ignored
    if (ragged && forced) {
        // CUTENSOR_AMD_H16_WAVES names the kernel: one of those that mask a partial K-tile (4x, 4m, 4m4, 4q), or the general family
        if (!usable || (var != 48 && var != 56 && var != 64 && var != 80)) return false;
        split = auto_split(var);
    } else if (ragged) {
        // the kernels of the family that mask a partial K-tile (the RAG instantiations): the 256 x 256 four-wave kernel, the 128 x 128
        // pair and the 64 x 64 tile — every candidate of the planner but the persistent kernel
        double best = 1e30;
        for (int cand : {48, 64, 56, 80}) {
            if (layoutIdx + cand >= count) continue;
            const uint64_t as = auto_split(cand);
            for (uint64_t sp : {(uint64_t)1, as}) {
                const double t = model_us(cand, sp);
                if (t < best) { best = t; var = cand; split = sp; c.stripKernel = -1; }
                if (as == 1) break;
            }
            uint32_t mi = 0, ni = 0;
            const double ts = stripsOK ? strip_us(cand, mi, ni) : 1e30;
            if (ts < 0.97 * best) { best = ts; var = cand; split = 1; c.stripKernel = layoutIdx + 80; c.mInt = mi; c.nInt = ni; }
        }
    } else if (forced && usable) {
        split = auto_split(var);
    } else {
        // the planner's own choice: the 256 x 256 family (four-wave 16x16x32 kernel; eight-wave kernel for short K ranges, below), the
        // 128 x 128 mid-size family and the 64 x 64 tile, each without split-K and at its automatic split
        double best = 1e30;
        static const bool noPersistent = [] { const char* e = CTAMD_HOOK_ENV("CUTENSOR_AMD_H16P"); return e && e[0] == '0'; }();
        // The persistent kernel earns its place by streaming interior tiles into each other, which needs the epilogue that stays out of the
        // operand ring (gett_h16p.hip, curOK): one M and one N mode, 16-byte lanes in D (batch modes stream since round 6).  Tiles that cannot stream are set up
        // serially behind the previous epilogue and the kernel is SLOWER than the one-tile kernel then (measured with beta != 0, the one
        // condition only the call knows: 8192^3 1377 against 1429-1435 TFLOP/s, 8192^2 x 1024 692 against 819, x 512 432 against 526,
        // profiles/r05r_h16p_beta.jsonl — cutensorContract launches the one-tile twin for beta != 0, api.cpp).
        // ... and the hand-over of tile i's last K-tile bodies to tile i + 1 needs an even K-tile count (gett_h16p.hip, switchAt; two
        // K-tiles per tile stream since round 6).  An odd count is made even by a zero K-tile (the odometer closes the descriptors once the
        // K range is exhausted: no memory access, zeros in LDS, 2270 cycles of MFMAs on zeros) — worth it while that is a small price for
        // the 7.8k-cycle prologue it saves per tile: up to 15 K-tiles (K = 64: attention scores with 64-wide heads)
        const bool streamable = v.M.size() == 1 && v.N.size() == 1 && v.N[0].sD == 1 && v.N[0].extent % 8 == 0 &&
                                v.M[0].sD % 8 == 0 && v.alignD % 16 == 0 && (kTiles % 2 == 0 || kTiles <= 15);
        for (int cand : {48, 88, 64, 56, 80}) {
            if (cand == 88 && (noPersistent || !streamable)) continue;
            if (layoutIdx + cand >= count) continue;
            const uint64_t as = auto_split(cand);
            for (uint64_t sp : {(uint64_t)1, as}) {
                const double t = model_us(cand, sp);
                if (t < best) { best = t; var = cand; split = sp; c.stripKernel = -1; }
                if (as == 1) break;
            }
            uint32_t mi = 0, ni = 0;
            const double ts = stripsOK ? strip_us(cand, mi, ni) : 1e30;
            if (ts < 0.97 * best) { best = ts; var = cand; split = 1; c.stripKernel = layoutIdx + 80; c.mInt = mi; c.nInt = ni; }
        }
    }
    c.kernel = layoutIdx + var;
    if (const char* fs = ctamd_research_env("CUTENSOR_AMD_H16_SPLITK")) {   // measurement knob: this many slices (if the workspace allows)
        const uint64_t want = std::strtoull(fs, nullptr, 10);
        if (want >= 1 && want <= kTiles && want * perSliceBytes <= std::max<uint64_t>(wsLimit, 1)) split = want;
        if (want == 1) split = 1;
    }
    const uint64_t tilesPerSlice = (kTiles + split - 1) / split;
    c.splitK = (uint32_t)((kTiles + tilesPerSlice - 1) / tilesPerSlice);
    c.kPerSlice = (uint32_t)(tilesPerSlice * 64);
    c.workspace = (c.splitK > 1) ? (uint64_t)c.splitK * perSliceBytes : 0ull;
    c.estimateUs = model_us(var, c.splitK);
    if (c.stripKernel >= 0) { uint32_t mi = 0, ni = 0; c.estimateUs = strip_us(var, mi, ni); }
    // (Rounds 2-3 sent K ranges of at most 16 K-tiles to the eight-wave kernel, whose fixed cost per workgroup was ~4 us lower; with the
    // shorter prologue and the pipelined epilogue of round 4 the four-wave kernel is ahead there too: 8192^2 x 256 / 512 / 1024
    // 71.6 / 96.4 / 148 us against 72.7 / 101 / 157, profiles/r04z_short_k_4x_vs_8.txt.)
    return true;
}

std::vector<ContractionChoice> rank_h16_choices(const ContractionView& v, uint64_t wsLimit, int numCUs) {
    std::vector<ContractionChoice> out;
    ContractionChoice base;
    if (!pick_h16_choice(v, wsLimit, numCUs, base)) return out;
    out.push_back(base);
    if (h16_needs_rag(v)) return out;              // ragged K / partial units: the planner's pick among the kernels that mask (pick_h16_choice)
    int count = 0;
    const int layoutIdx = base.kernel % 8, variant = base.kernel - layoutIdx;
    const GettKernelInfo* tab = gett_h16_kernels(&count);
    for (int other : {0, 48, 88, 56, 64, 72, 80, 40, 32, 16, 24, 8}) {  // ping-pong rows, four waves register-staged, streamed (free-running waves), four waves streamed, four waves
        if (other == variant || layoutIdx + other >= count) continue;
        if (tab[layoutIdx + other].ablation == 2) continue;        // a retired family, not built into this library (research builds only)
        ContractionChoice c = base;
        c.kernel = layoutIdx + other;
        c.stripKernel = -1;                                        // the whole grid on this kernel
        c.mInt = c.nInt = 0;
        out.push_back(c);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// General MFMA family (kernels/gett_gen.inc): bf16 / fp16 shapes the aligned kernels refuse, fp64, complex64 / complex128.
// Per operand: the orientation of its staged units (along k when the fastest contracted mode is its stride-1 mode, along rows
// when the fastest free mode is) and the widest unit V its extents, strides and pointer alignment admit; the kernel runs both
// operands at the smaller V.  Tile: the large one when it still fills the chip.  Split-K for every element type when the output
// tiles alone leave most CUs idle: partials in the accumulator type — fp32 for 16-bit data (folded by launch_splitk_reduce with one
// rounding), double / float2 / double2 otherwise (launch_gen_splitk_reduce).
// ---------------------------------------------------------------------------------------------
static int gen_elem_of(hipDataType t) {
    switch (t) {
        case HIP_R_16BF: return GEN_BF16;
        case HIP_R_16F:  return GEN_F16;
        case HIP_R_64F:  return GEN_F64;
        case HIP_C_32F:  return GEN_C32;
        case HIP_C_64F:  return GEN_C64;
        default:         return -1;
    }
}

// widest unit (elements) operand `slotA` admits with units along k (orient 1) or along rows (orient 0); 1 = element gathers
static int gen_operand_vec(const ContractionView& v, bool slotA, int orient, int maxV) {
    const std::vector<CanonMode>& freeG = slotA ? v.M : v.N;
    const std::vector<CanonMode>& lead = orient ? v.K : freeG;
    if (lead.empty()) return 1;
    const CanonMode& m0 = lead.front();
    if ((slotA ? m0.sA : m0.sB) != 1) return 1;
    const int64_t es = (int64_t)dtype_size(v.dtype);
    const uint32_t align = slotA ? v.alignA : v.alignB;
    for (int V = maxV; V > 1; V >>= 1) {
        if (m0.extent % V != 0 || align % (uint32_t)(V * es) != 0) continue;
        bool ok = true;
        for (const std::vector<CanonMode>* g : {&freeG, &v.K, &v.L})
            for (const CanonMode& m : *g) {
                if (&m == &m0) continue;
                if ((slotA ? m.sA : m.sB) % V != 0) ok = false;
            }
        if (ok) return V;
    }
    return 1;
}

bool pick_gen_choice(const ContractionView& v, uint64_t wsLimit, int numCUs, ContractionChoice& c) {
    const int elem = gen_elem_of(v.dtype);
    if (elem < 0 || v.wide) return false;
    const int maxV = (elem == GEN_C64) ? 1 : (elem == GEN_F64 || elem == GEN_C32) ? 2 : 8;
    int orient[2], vec[2];
    for (int o = 0; o < 2; ++o) {
        const bool slotA = o == 0;
        const int vK = gen_operand_vec(v, slotA, 1, maxV), vF = gen_operand_vec(v, slotA, 0, maxV);
        if (vK > 1 && vK >= vF) { orient[o] = 1; vec[o] = vK; }
        else if (vF > 1)        { orient[o] = 0; vec[o] = vF; }
        else {
            // element gathers: neighbouring lanes along whichever direction has the smaller stride
            const std::vector<CanonMode>& freeG = slotA ? v.M : v.N;
            const int64_t sK = v.K.empty() ? INT64_MAX : std::llabs(slotA ? v.K.front().sA : v.K.front().sB);
            const int64_t sF = freeG.empty() ? INT64_MAX : std::llabs(slotA ? freeG.front().sA : freeG.front().sB);
            orient[o] = (sK <= sF) ? 1 : 0;
            vec[o] = 1;
        }
    }
    int V = std::min(vec[0], vec[1]);
    if (elem <= GEN_F16 && V == 4) V = 2;          // instantiated widths: 8 / 2 / 1 (16-bit), 2 / 1 (fp64, complex64), 1 (complex128)
    int count = 0;
    const GettKernelInfo* tab = gett_gen_kernels(&count);
    // candidates of this (type, V, orientation pair): the table lists the larger tile first
    int big = -1, small = -1;
    for (int i = 0; i < count; ++i)
        if (tab[i].elem == elem && tab[i].vec == V && tab[i].layA == orient[0] && tab[i].layB == orient[1]) {
            if (big < 0) big = i;
            small = i;
        }
    if (big < 0) return false;
    auto tiles_of = [&](int k) {
        return std::ceil((double)v.totM / tab[k].bm) * std::ceil((double)v.totN / tab[k].bn) * (double)v.totL;
    };
    c = ContractionChoice{};
    c.family = 2;
    c.kernel = (tiles_of(big) >= 0.6 * numCUs) ? big : small;
    const GettKernelInfo& k = tab[c.kernel];
    const double tiles = tiles_of(c.kernel);
    const uint64_t kTiles = (v.totK + k.bk - 1) / k.bk;
    // split-K when the output tiles alone leave most CUs idle: partial tiles [slice][L][M][N] in the accumulator type
    // (fp32 for 16-bit data, double / float2 / double2 for fp64 / complex64 / complex128)
    const uint64_t accBytes = (elem <= GEN_F16) ? 4ull : (elem == GEN_C64) ? 16ull : 8ull;
    const uint64_t perSliceBytes = v.totL * v.totM * v.totN * accBytes;
    // From 16 K-tiles on, slices of at least four.  (Round 4 raised this to 48 K-tiles on a misread pair of numbers; the records say the
    // opposite — the reference's fp16 case 'mlik,lkjm->lij', 50 batches x 32 K-tiles on 2-byte gathers: 19.2 us in five slices
    // (profiles/r04a_bench_gen.jsonl, 250 workgroups), 32.1-32.6 us unsplit (the bench line of rounds 4-6, 50 workgroups on 256 CUs):
    // a K-tile of this family costs ~0.8 us, the fold's launch ~4.)
    uint64_t split = 1;
    if (tiles * 2.0 <= (double)numCUs && kTiles >= 16) {
        // (the 64 x 64 kernels run two workgroups per CU — tests/test_kernel_resources.py pins their 256 registers — so their slices
        // may number twice the idle CUs)
        const double slots = (double)numCUs * ((k.bm * k.bn <= 64 * 64) ? 2.0 : 1.0);
        split = std::min<uint64_t>((uint64_t)(slots / tiles), kTiles / 4);
        while (split > 1 && split * perSliceBytes > wsLimit) --split;
        if (split < 2) split = 1;
    }
    const uint64_t tilesPerSlice = (kTiles + split - 1) / split;
    c.splitK = (uint32_t)((kTiles + tilesPerSlice - 1) / tilesPerSlice);
    c.kPerSlice = (uint32_t)(tilesPerSlice * k.bk);
    c.workspace = (c.splitK > 1) ? (uint64_t)c.splitK * perSliceBytes : 0ull;
    // rough time: the family's MFMA rate for the type at ~50 % utilisation (only used for logs / describe)
    const double flopPerClkCU = (elem <= GEN_F16) ? 4096.0 : (elem == GEN_C32) ? 256.0 : 128.0;
    const double flops = ((elem >= GEN_C32) ? 8.0 : 2.0) * k.bm * k.bn * (double)c.kPerSlice;
    c.estimateUs = std::ceil(tiles * c.splitK / (double)numCUs) * flops / (flopPerClkCU * 2.4e9 * 0.5) * 1e6 + 2.0;
    return true;
}

static void fill_group(ModeGroup& g, const std::vector<CanonMode>& modes) {
    std::memset(&g, 0, sizeof(g));
    g.n = (int32_t)modes.size();
    uint64_t tot = 1;
    for (int i = 0; i < kMaxGroupModes; ++i) g.div[i] = FastDiv{1u, 0u, 0u};   // padding: digit = remaining index (0)
    for (size_t i = 0; i < modes.size(); ++i) {
        g.div[i] = make_fastdiv((uint32_t)modes[i].extent);
        tot *= (uint64_t)modes[i].extent;
    }
    g.total = (uint32_t)tot;
}

void fill_gett_params(const ContractionView& v, const ContractionChoice& c, GettParams& p,
                      SplitKReduceParams& r) {
    std::memset(&p, 0, sizeof(p));
    std::memset(&r, 0, sizeof(r));
    fill_group(p.gM, v.M);
    fill_group(p.gN, v.N);
    fill_group(p.gK, v.K);
    fill_group(p.gL, v.L);
    for (size_t i = 0; i < v.M.size(); ++i) {
        p.gM.stride[0][i] = v.M[i].sA; p.gM.stride[1][i] = v.M[i].sD; p.cStrideM[i] = v.M[i].sC;
    }
    for (size_t i = 0; i < v.N.size(); ++i) {
        p.gN.stride[0][i] = v.N[i].sB; p.gN.stride[1][i] = v.N[i].sD; p.cStrideN[i] = v.N[i].sC;
    }
    for (size_t i = 0; i < v.K.size(); ++i) {
        p.gK.stride[0][i] = v.K[i].sA; p.gK.stride[1][i] = v.K[i].sB;
    }
    for (size_t i = 0; i < v.L.size(); ++i) {
        p.gL.stride[0][i] = v.L[i].sA; p.gL.stride[1][i] = v.L[i].sB; p.gL.stride[2][i] = v.L[i].sD;
        p.cStrideL[i] = v.L[i].sC;
    }
    int count = 0;
    const GettKernelInfo* tab = (c.family == 2) ? gett_gen_kernels(&count) : (c.family == 1) ? gett_h16_kernels(&count) : gett_f32_kernels(&count);
    int bm = 16, bn = 16, bk = 16;
    if (c.kernel >= 0 && c.kernel < count) { bm = tab[c.kernel].bm; bn = tab[c.kernel].bn; bk = tab[c.kernel].bk; }
    (void)bk;
    p.tilesM = (uint32_t)((v.totM + bm - 1) / bm);
    p.tilesN = (uint32_t)((v.totN + bn - 1) / bn);
    p.splitK = c.splitK;
    p.kPerSlice = c.kPerSlice ? c.kPerSlice : (uint32_t)v.totK;
    p.nBlocks = (uint32_t)((uint64_t)p.tilesM * p.tilesN * p.splitK * v.totL);

    {   // bytes from an operand's first element to one past its last (cutensorContract adds the pointer: GettParams::endA / endB)
        const uint64_t es = (uint64_t)dtype_size(v.dtype);
        uint64_t spanA = 1, spanB = 1;
        for (const std::vector<CanonMode>* g : {&v.L, &v.M, &v.K})
            for (const CanonMode& m : *g) spanA += (uint64_t)(m.extent - 1) * (uint64_t)std::llabs(m.sA);
        for (const std::vector<CanonMode>* g : {&v.L, &v.N, &v.K})
            for (const CanonMode& m : *g) spanB += (uint64_t)(m.extent - 1) * (uint64_t)std::llabs(m.sB);
        p.endA = spanA * es;
        p.endB = spanB * es;
        p.ragged = (c.family == 1 && h16_needs_rag(v)) ? 1u : 0u;
        if (c.family == 1 && h16_sweep_ragged(v)) p.ragged |= 2u | ((uint32_t)h16_k_tiles(v) << 2);   // sweep-ragged K: the padded K-tile count
        if (c.family == 0 && v.dtype == HIP_R_32F && c.kernel >= 0) {
            int cnt = 0;
            const GettKernelInfo* t32 = gett_f32_kernels(&cnt);
            if (c.kernel < cnt && t32[c.kernel].fragPartials && f32_needs_rag(v)) p.ragged = 1u;
        }
    }

    r.gM = p.gM; r.gN = p.gN; r.gL = p.gL;
    std::memcpy(r.cStrideM, p.cStrideM, sizeof(r.cStrideM));
    std::memcpy(r.cStrideN, p.cStrideN, sizeof(r.cStrideN));
    std::memcpy(r.cStrideL, p.cStrideL, sizeof(r.cStrideL));
    r.splitK = c.splitK;
    r.tilesM = p.tilesM; r.tilesN = p.tilesN;
    r.fragTM = (uint32_t)bm / 32u; r.fragTN = (uint32_t)bn / 32u;
    // output type of launch_splitk_reduce (0 fp32, 1 bf16, 2 fp16): set for 16-bit data only — fp64 / complex partials of the general
    // family are folded by launch_gen_splitk_reduce, which takes the element type from the kernel table and never reads this field
    r.outType = (v.dtype == HIP_R_16BF) ? 1 : (v.dtype == HIP_R_16F) ? 2 : 0;
}

}  // namespace ctamd
