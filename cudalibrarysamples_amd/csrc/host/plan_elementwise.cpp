// plan_elementwise.cpp — host planners for cutensorCreatePermutation / ElementwiseBinary /
// cutensorCreateReduction (reference call sites: cuTENSOR/elementwise_permute.cu:142-149,
// cuTENSOR/elementwise_binary.cu:149-153, cuTENSOR/reduction.cu:141-146, cuTENSOR/einsum.cu:346-351).
//
// Both planners canonicalise the strided N-mode problem (drop extent-1 modes, sort by stride, fuse
// jointly contiguous neighbours) and then choose the kernel variant whose vector-width and alignment
// preconditions hold (see kernels/elementwise.hip and kernels/reduce.hip).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "internal.hpp"
#include "api_guard.hpp"

namespace ctamd {

namespace {

struct EwMode {
    int64_t extent;
    int64_t sA = 0, sD = 0, sC = 0;
    int64_t sX = 0;   // second permuted operand (element-wise trinary), 0 when unused
};

int find_label(const std::vector<int32_t>& modes, int32_t l) {
    for (size_t i = 0; i < modes.size(); ++i)
        if (modes[i] == l) return (int)i;
    return -1;
}

bool dup_labels(const std::vector<int32_t>& modes) {
    for (size_t i = 0; i < modes.size(); ++i)
        for (size_t j = i + 1; j < modes.size(); ++j)
            if (modes[i] == modes[j]) return true;
    return false;
}

void fuse(std::vector<EwMode>& g, bool useC) {
    std::vector<EwMode> out;
    for (const EwMode& m : g) {
        if (!out.empty()) {
            EwMode& p = out.back();
            bool ok = (m.sA == p.sA * p.extent) && (m.sD == p.sD * p.extent) && (m.sX == p.sX * p.extent);
            if (useC) ok = ok && (m.sC == p.sC * p.extent);
            if (ok && p.extent * m.extent < (1ll << 31)) {
                p.extent *= m.extent;
                continue;
            }
        }
        out.push_back(m);
    }
    g.swap(out);
}

bool fill_rest(ModeGroup& g, const std::vector<EwMode>& modes) {
    std::memset(&g, 0, sizeof(g));
    if ((int)modes.size() > kMaxGroupModes) return false;
    g.n = (int32_t)modes.size();
    uint64_t tot = 1;
    for (int i = 0; i < kMaxGroupModes; ++i) g.div[i] = FastDiv{1u, 0u, 0u};   // padding modes
    for (size_t i = 0; i < modes.size(); ++i) {
        g.div[i] = make_fastdiv((uint32_t)modes[i].extent);
        g.stride[0][i] = modes[i].sA;
        g.stride[1][i] = modes[i].sD;
        g.stride[2][i] = modes[i].sC;
        tot *= (uint64_t)modes[i].extent;
        if (tot >= (1ull << 31)) return false;
    }
    g.total = (uint32_t)tot;
    return true;
}

}  // namespace

cutensorStatus_t plan_elementwise(const cutensorOperationDescriptor& op, EwPlan& plan, std::string* why, bool allowWide) {
    auto fail = [&](cutensorStatus_t st, const char* msg) {
        if (why) *why = msg;
        return st;
    };
    const TensorUse &A = op.A, &C = op.C, &D = op.D;
    const bool usesC = C.present;
    // second permuted operand: only the trinary planner passes a descriptor of kind ElementwiseBinary with B present
    const bool usesX = op.kind == OpKind::ElementwiseBinary && op.B.present;
    if (usesX && (op.B.desc.dtype != D.desc.dtype || dup_labels(op.B.modes))) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "second permuted operand");
    // complex data (python/einsum.h:326-343 routes a unary einsum on complex tensors here through cutensorCreateReduction):
    // one permuted operand, combiner ADD or MUL, conjugation of A / C — the (re, im)-pair form of the generic kernel
    const bool cplx = D.desc.dtype == HIP_C_32F || D.desc.dtype == HIP_C_64F;
    if (cplx && usesX) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "complex element-wise trinary operations");
    if (cplx && op.kind == OpKind::ElementwiseBinary && op.opReduce != CUTENSOR_OP_ADD && op.opReduce != CUTENSOR_OP_MUL)
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "MAX / MIN are not defined on complex data");
    if (dup_labels(A.modes) || dup_labels(D.modes) || (usesC && dup_labels(C.modes)))
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "repeated mode label inside one tensor");
    if (A.desc.dtype != D.desc.dtype || (usesC && C.desc.dtype != D.desc.dtype))
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "mixed data types");
    // unary operators: IDENTITY, and CONJ (a no-op on real data)
    auto unary_ok = [](cutensorOperator_t o) { return o == CUTENSOR_OP_IDENTITY || o == CUTENSOR_OP_CONJ; };
    if (!unary_ok(A.op) || (usesC && !unary_ok(C.op)) || (usesX && op.B.op != CUTENSOR_OP_IDENTITY))
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "only the identity and conjugation operators are implemented");
    for (int32_t l : A.modes)
        if (find_label(D.modes, l) < 0 && A.desc.extent[find_label(A.modes, l)] != 1)
            return fail(CUTENSOR_STATUS_INVALID_VALUE, "mode of A missing from the output");
    if (usesC)
        for (int32_t l : C.modes)
            if (find_label(D.modes, l) < 0 && C.desc.extent[find_label(C.modes, l)] != 1)
                return fail(CUTENSOR_STATUS_INVALID_VALUE, "mode of C missing from the output");

    std::vector<EwMode> modes;
    for (size_t i = 0; i < D.modes.size(); ++i) {
        EwMode m;
        m.extent = D.desc.extent[i];
        m.sD = D.desc.stride[i];
        const int ia = find_label(A.modes, D.modes[i]);
        if (ia >= 0) {
            if (A.desc.extent[ia] != m.extent) return fail(CUTENSOR_STATUS_INVALID_VALUE, "extent mismatch between A and output");
            m.sA = A.desc.stride[ia];
        }   // absent => stride 0 (broadcast)
        if (usesX) {
            const int ix = find_label(op.B.modes, D.modes[i]);
            if (ix >= 0) {
                if (op.B.desc.extent[ix] != m.extent) return fail(CUTENSOR_STATUS_INVALID_VALUE, "extent mismatch between B and output");
                m.sX = op.B.desc.stride[ix];
            }
        }
        if (usesC) {
            const int ic = find_label(C.modes, D.modes[i]);
            if (ic >= 0) {
                if (C.desc.extent[ic] != m.extent) return fail(CUTENSOR_STATUS_INVALID_VALUE, "extent mismatch between C and output");
                m.sC = C.desc.stride[ic];
            }
        }
        if (m.extent <= 0) return fail(CUTENSOR_STATUS_INVALID_VALUE, "non-positive extent");
        if (m.extent == 1) continue;
        modes.push_back(m);
    }
    std::stable_sort(modes.begin(), modes.end(), [](const EwMode& x, const EwMode& y) { return x.sD < y.sD; });
    fuse(modes, usesC);
    // Everything fused into ONE contiguous mode (identical layouts: a flat copy / axpby, e.g. the packed identity permutations of
    // cutensorMp, 'ij->ij', elementwise_binary.cu with equal mode orders): as a single row the row-copy kernel gets one 1-KiB segment per
    // workgroup and 7 of its 8 tile rows idle — 1.85 TB/s on 96 MB (profiles/r06z_trinary_parts.jsonl).  Cut the row into rows of d
    // elements, d the largest divisor of the extent in [256, 4096] that keeps 16-byte lanes: the same bytes as a 2-D row copy (round 6).
    if (modes.size() == 1 && modes[0].sD == 1 && modes[0].sA == 1 && (!usesC || modes[0].sC == 1) && (!usesX || modes[0].sX == 1) &&
        modes[0].extent >= 2 * 4096) {
        const int64_t E = modes[0].extent;
        int64_t d = 0;
        for (int64_t cand = 4096; cand >= 256; cand -= 16)
            if (E % cand == 0) { d = cand; break; }
        if (d > 0) {
            EwMode lo = modes[0], hi = modes[0];
            lo.extent = d;
            hi.extent = E / d;
            hi.sA = hi.sD = d;
            hi.sC = usesC ? d : 0;
            hi.sX = usesX ? d : 0;
            modes.clear();
            modes.push_back(lo);
            modes.push_back(hi);
        }
    }

    plan = EwPlan{};
    plan.usesC = usesC;
    Ew2DParams& p = plan.p;
    std::memset(&p, 0, sizeof(p));
    p.E0 = p.E1 = 1;
    if (op.kind == OpKind::ElementwiseBinary) p.opAC = (int32_t)op.opReduce;   // ADD / MUL / MAX / MIN; every other caller adds
    p.conjA = (cplx && A.op == CUTENSOR_OP_CONJ) ? 1 : 0;
    p.conjC = (cplx && usesC && C.op == CUTENSOR_OP_CONJ) ? 1 : 0;
    std::vector<EwMode> rest;
    int i1 = -1;
    if (!modes.empty()) {
        const EwMode& m0 = modes[0];
        if (m0.extent >= (1ll << 31)) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "mode extent >= 2^31");
        p.E0 = (uint32_t)m0.extent; p.sA0 = m0.sA; p.sD0 = m0.sD; p.sC0 = m0.sC; p.sX0 = m0.sX;
        if (modes.size() > 1) {
            i1 = 1;
            if (m0.sA != 1) {   // partner dim = A's fastest remaining mode
                for (size_t i = 1; i < modes.size(); ++i)
                    if (modes[i].sA != 0 && (modes[i1].sA == 0 || modes[i].sA < modes[i1].sA)) i1 = (int)i;
            }
            const EwMode& m1 = modes[i1];
            p.E1 = (uint32_t)m1.extent; p.sA1 = m1.sA; p.sD1 = m1.sD; p.sC1 = m1.sC; p.sX1 = m1.sX;
        }
        for (size_t i = 1; i < modes.size(); ++i)
            if ((int)i != i1) rest.push_back(modes[i]);
    }
    if (!fill_rest(p.rest, rest)) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "too many unfusable modes");
    for (size_t i = 0; i < rest.size(); ++i) p.restX[i] = rest[i].sX;

    // ---- variant --------------------------------------------------------------------------
    // vector paths: fp32 (4-element lanes) and bf16 / fp16 (8-element lanes); "mult4" = multiple of the lane width
    const bool h16 = D.desc.dtype == HIP_R_16BF || D.desc.dtype == HIP_R_16F;
    // 8- / 16-byte elements (round 6; elementwise.hip ew_transpose_wide_kernel / ew_rowcopy_wide_kernel): permutations and binary
    // operations on fp64 / complex64 / complex128 — a 16-byte lane holds 2 / 2 / 1 elements; never with a trinary operand (E / X)
    const bool wide = allowWide && !usesX && (D.desc.dtype == HIP_R_64F || cplx);
    const bool f32 = D.desc.dtype == HIP_R_32F || (h16 && !usesX) || wide;
    const int64_t vec = wide ? 16 / (int64_t)dtype_size(D.desc.dtype) : h16 ? 8 : 4;
    const bool aligned = (A.desc.alignment % 16 == 0) && (D.desc.alignment % 16 == 0) &&
                         (!usesC || C.desc.alignment % 16 == 0);
    auto mult4 = [vec](int64_t s) { return s % vec == 0; };
    bool restOK = true;
    for (const EwMode& m : rest) restOK = restOK && mult4(m.sA) && mult4(m.sD) && (!usesC || mult4(m.sC));
    // C is read with 16-byte lanes only when it is contiguous along dim0; otherwise element-wise
    const bool cOK = !usesC || p.sC0 != 1 || (mult4(p.sC1));
    plan.variant = EW_GENERIC;
    int t0 = 64, t1 = 4;
    bool xOK = true;   // the transposing variant reads X exactly like A: 16-byte lanes along dim1
    if (usesX) {
        xOK = op.B.desc.alignment % 16 == 0 && p.sX1 == 1 && mult4(p.sX0);
        for (const EwMode& m : rest) xOK = xOK && mult4(m.sX);
    }
    if (f32 && aligned && restOK && cOK && p.sD0 == 1 && p.E0 % vec == 0) {
        if (i1 >= 0 && p.sA1 == 1 && p.sA0 != 1 && p.E1 % vec == 0 && mult4(p.sA0) && mult4(p.sD1) && xOK) {
            plan.variant = EW_TRANSPOSE; t0 = 64; t1 = 64;
            // fp32: the widest written row segment the extent fills (elementwise.hip: 256-B -> 512-B -> 1-KiB segments are
            // 6.14 -> 6.45 -> 6.56 TB/s at 2048^3); the two-tile (X) form stops at 128 to keep its LDS footprint
            if (D.desc.dtype == HIP_R_32F) {
                const int64_t widest = usesX ? 128 : 256;
                for (int64_t cand = widest; cand > 64; cand /= 2)
                    if (p.E0 % cand == 0 || p.E0 >= 4 * cand) { t0 = (int)cand; break; }
            } else if (wide) {
                // fp64 / complex64: 128 x 32 elements (1-KiB written, 256-B read segments); complex128: 64 x 32 (1 KiB / 512 B)
                t0 = D.desc.dtype == HIP_C_64F ? 64 : 128;
                t1 = 32;
            } else if (h16 && p.E1 % 64 == 0) {
                // 16-bit: the wide kernel handles full tiles only (T0 x T1 elements: 512-B / 256-B written, 256-B / 128-B read segments)
                if (p.E0 % 256 == 0) t0 = 256;
                else if (p.E0 % 128 == 0) t0 = 128;
                if (t0 > 64) {
                    t1 = (p.E1 % 128 == 0) ? 128 : 64;
                    if (const char* e = ctamd_research_env("CUTENSOR_AMD_H16_TRANSPOSE_T1")) { const int v = std::atoi(e); if ((v == 64 || v == 128) && p.E1 % v == 0) t1 = v; }
                }
            }
        } else if (!usesX && p.sA0 == 1 && mult4(p.sA1) && mult4(p.sD1)) {
            plan.variant = EW_ROWCOPY; t0 = wide ? 64 * (int)vec : h16 ? 512 : 256; t1 = 8;
        }
    }
    // EW_BLOCK (round 6): a pure permutation of 2- / 4-byte elements that the tiled kernels refuse, whose first n modes in D's order are
    // packed in D AND the same set of modes is packed at the front of A — contiguous blocks, permuted inside (elementwise.hip
    // ew_block_kernel).  The smallest such n; blocks of at most 32 KiB; a workgroup takes as many blocks as fill ~4 Ki elements.
    // Also preferred to the transposing kernel when its tiles would be mostly padding (dim0 = b = 8, dim1 = d = 40 of a 64 x 64 tile:
    // [d = 40, c, b | a] -> [b, c, d | a] at 0.7-0.9 TB/s, profiles/r06zze_block_permute.jsonl).
    const double tileFill = (double)p.E0 * (double)p.E1 / ((double)((p.E0 + t0 - 1) / t0) * t0 * (double)((p.E1 + t1 - 1) / t1) * t1);
    if ((plan.variant == EW_GENERIC || (plan.variant == EW_TRANSPOSE && tileFill < 0.5)) && op.kind == OpKind::Permutation && !usesC && !usesX && !cplx && (h16 || D.desc.dtype == HIP_R_32F) &&
        op.padLeft.empty() && op.padRight.empty() && modes.size() >= 2) {
        const uint64_t cap = 32768 / (uint64_t)dtype_size(D.desc.dtype);
        for (size_t n = 2; n <= modes.size() && n <= 4; ++n) {
            uint64_t run = 1;
            bool ok = true;
            for (size_t i = 0; i < n && ok; ++i) { ok = modes[i].sD == (int64_t)run; run *= (uint64_t)modes[i].extent; }
            if (!ok || run > cap) break;                                   // (D's leading modes are not packed, or the block outgrew LDS: no larger n helps)
            std::vector<size_t> byA(n);
            for (size_t i = 0; i < n; ++i) byA[i] = i;
            std::sort(byA.begin(), byA.end(), [&](size_t x, size_t y) { return modes[x].sA < modes[y].sA; });
            uint64_t runA = 1;
            for (size_t i = 0; i < n && ok; ++i) { ok = modes[byA[i]].sA == (int64_t)runA; runA *= (uint64_t)modes[byA[i]].extent; }
            if (!ok) continue;
            std::vector<EwMode> others(modes.begin() + (long)n, modes.end());
            if (!fill_rest(p.blkRest, others)) break;
            p.blkN = (uint32_t)n;
            p.blkTotal = (uint32_t)run;
            for (size_t i = 0; i < n; ++i) { p.blkDiv[i] = make_fastdiv((uint32_t)modes[i].extent); p.blkSrc[i] = (uint32_t)modes[i].sA; }
            for (size_t i = n; i < 4; ++i) { p.blkDiv[i] = make_fastdiv(1u); p.blkSrc[i] = 0u; }
            uint64_t group = std::max<uint64_t>(1, 4096 / run);
            group = std::min<uint64_t>(group, cap / run);
            group = std::min<uint64_t>(group, std::max<uint64_t>(1, (uint64_t)p.blkRest.total / 512));   // (keep at least ~512 workgroups)
            p.blkGroup = (uint32_t)std::max<uint64_t>(1, group);
            p.blkBlocks = (p.blkRest.total + p.blkGroup - 1) / p.blkGroup;
            {   // 16-byte lanes: block size and the other modes' strides multiples of the lane's elements (the base pointers are checked at launch)
                const uint64_t lane = 16 / (uint64_t)dtype_size(D.desc.dtype);
                bool v16 = run % lane == 0;
                for (const EwMode& m : others) v16 = v16 && m.sA % (int64_t)lane == 0 && m.sD % (int64_t)lane == 0;
                p.blkVec = v16 ? 1u : 0u;
            }
            plan.variant = EW_BLOCK;
            break;
        }
    }
    // EW_TRANSPOSE_ANY (round 6): what is left for the element-gather kernel although it IS a transposition — D contiguous along dim0, A
    // along dim1, but odd extents / strides / base alignment (4097 x 4099: 0.9-1.5 TB/s there, each lane of a load on another line)
    const bool anyForced = CTAMD_HOOK_ENV("CUTENSOR_AMD_EW_ANY") && CTAMD_HOOK_ENV("CUTENSOR_AMD_EW_ANY")[0] == '1';   // measurement: also where the 16-byte-lane kernel applies
    // ... and it is the faster kernel for MID-SIZE pure permutations the 16-byte-lane kernels do take: their wide tiles (fp32: 256 x 64)
    // leave a 4096^2 transposition on 1024 workgroups — 4.6 TB/s against 5.9 on 4096 tiles of 64 x 64 — and rows whose pitch is no
    // multiple of 128 bytes cost them more (4104^2: 3.4 / 5.2; bf16 on the narrow 64 x 64 kernel 2.6 / 3.4).  Large tensors stay
    // (1024^3: 5.6 / 5.0 fp32, 5.9 / 4.3 bf16 on the wide kernel) — profiles/r06zzo_permute_any_ab.jsonl.
    // Second A/B, 3-D reversals of 96 MB .. 1.2 GB (profiles/r06zzq_permute_mid_ab.jsonl): fp32 (400, 200, 300) 3.6 / 4.7, (512, 256, 256)
    // 5.1 / 6.4, (800, 400, 300) 4.0 / 5.2, 640^3 6.2 / 5.9; bf16 on the narrow kernel (1000, 500, 600) 2.3 / 3.0.  So: by bytes —
    // fp32 below 512 MB, the narrow 16-bit kernel below 1 GB.
    const uint64_t vecTiles = (uint64_t)((p.E0 + t0 - 1) / t0) * (uint64_t)((p.E1 + t1 - 1) / t1) * (uint64_t)p.rest.total;
    const uint64_t tensorBytes = (uint64_t)p.E0 * (uint64_t)p.E1 * (uint64_t)p.rest.total * (uint64_t)dtype_size(D.desc.dtype);
    const bool anyOff = CTAMD_HOOK_ENV("CUTENSOR_AMD_EW_ANY") && CTAMD_HOOK_ENV("CUTENSOR_AMD_EW_ANY")[0] == '0';      // tests: the 16-byte-lane kernels on small tensors
    const bool midSize = !anyOff && plan.variant == EW_TRANSPOSE && ((D.desc.dtype == HIP_R_32F && (vecTiles < 4096 || tensorBytes < (512ull << 20))) ||
                                                                     (h16 && t0 == 64 && (vecTiles < 16384 || tensorBytes < (1ull << 30))));
    // (the binary form of cutensorElementwiseBinaryExecute too — allowWide tells it from the trinary planner's passes, which attach E / X
    // operands at run time: C joins in the store phase, an instantiation of its own)
    const bool anyKind = (op.kind == OpKind::Permutation && !usesC) || (op.kind == OpKind::ElementwiseBinary && allowWide && usesC);
    // (measured, profiles/r06zzu_binary_any_ab.jsonl: the binary form gains where it came from the element-gather kernel — odd extents — and
    // LOSES to the lane kernels at mid size, 4.14 / 3.71 TB/s at the sample's (400, 200, 300): the mid-size rule is the permutation's alone)
    const bool fromLanes = (midSize || (anyForced && plan.variant == EW_TRANSPOSE)) && op.kind == OpKind::Permutation;
    if ((plan.variant == EW_GENERIC || fromLanes) && anyKind && !usesX && !cplx && (h16 || D.desc.dtype == HIP_R_32F) &&
        op.padLeft.empty() && op.padRight.empty() && i1 >= 0 && p.sD0 == 1 && p.sA1 == 1 && p.sA0 != 1 && p.E0 >= 16 && p.E1 >= 16) {
        plan.variant = EW_TRANSPOSE_ANY;
        t0 = 64; t1 = 64;
        // 16-bit pure permutations with even extents, even strides and 4-byte-aligned bases: pairs per lane, 128 x 128 tiles
        // (ew_transpose_any_pair_kernel)
        if (h16 && op.kind == OpKind::Permutation && !usesC && p.E0 % 2 == 0 && p.E1 % 2 == 0 && p.sA0 % 2 == 0 && p.sD1 % 2 == 0 &&
            A.desc.alignment % 4 == 0 && D.desc.alignment % 4 == 0 && p.E0 >= 128 && p.E1 >= 128) {
            bool even = true;
            for (const EwMode& m : rest) even = even && m.sA % 2 == 0 && m.sD % 2 == 0;
            if (even) { t0 = 128; t1 = 128; }
        }
    }
    plan.usesX = usesX;
    p.tile0 = (uint32_t)t0;
    p.tile1 = (uint32_t)t1;
    p.tiles0 = (p.E0 + t0 - 1) / t0;
    p.tiles1 = (p.E1 + t1 - 1) / t1;
    p.divTiles0 = make_fastdiv(p.tiles0);
    p.divTiles1 = make_fastdiv(p.tiles1);
    const uint64_t nb = (uint64_t)p.tiles0 * p.tiles1 * p.rest.total;
    if (nb >= (1ull << 31)) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "tensor too large for the tile index space");
    p.nBlocks = (uint32_t)nb;
    // Tile order of the fp32 transposing kernel: when the tile's rows lie >= 1 MiB apart on BOTH sides (A's dim0 pitch and
    // D's dim1 pitch) and there is a rest index to walk instead, go rest-first with one contiguous eighth per XCD
    p.order = 0;
    p.idsPerXcd = (uint32_t)((nb + 7) / 8);
    p.divRest = make_fastdiv(p.rest.total);
    const int64_t esz = wide ? (int64_t)dtype_size(D.desc.dtype) : h16 ? 2 : 4;
    if (plan.variant == EW_TRANSPOSE && (D.desc.dtype == HIP_R_32F || wide || (h16 && t0 > 64)) && p.rest.total >= 64 && nb >= 4096 &&
        p.sA0 * esz >= (1 << 20) && p.sD1 * esz >= (1 << 20))
        p.order = 1;
    return CUTENSOR_STATUS_SUCCESS;
}

// D = opABC(opAB(alpha A, beta B), gamma C)  (cuTENSOR/elementwise_trinary.cu:174-182; the sample's
// D_{a,b,c} = alpha A_{c,b,a} + beta B_{c,a,b} + gamma C_{a,b,c}, :51-53).  The element-wise kernels take
// one permuted operand through their tile path plus operands that are walked with D's own strides, so:
//   * an operand of A / B that already has D's layout rides along as E: one pass, 4 |D| bytes;
//   * otherwise pass 1 writes D = alpha perm(A) and pass 2 combines in place with beta perm(B) and gamma C:
//     6 |D| bytes (the combiners ADD / MUL / MAX / MIN commute, so the order of A and B is free).
cutensorStatus_t plan_elementwise_trinary(const cutensorOperationDescriptor& op, EwTrinaryPlan& plan, std::string* why) {
    auto fail = [&](cutensorStatus_t st, const char* msg) {
        if (why) *why = msg;
        return st;
    };
    auto ok_op = [](cutensorOperator_t o) { return o == CUTENSOR_OP_ADD || o == CUTENSOR_OP_MUL || o == CUTENSOR_OP_MAX || o == CUTENSOR_OP_MIN; };
    if (!ok_op(op.opAB) || !ok_op(op.opReduce)) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "binary operator");
    if (op.A.op != CUTENSOR_OP_IDENTITY || op.B.op != CUTENSOR_OP_IDENTITY || op.C.op != CUTENSOR_OP_IDENTITY)
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "only the identity operator is implemented");
    if (op.D.desc.dtype == HIP_C_32F || op.D.desc.dtype == HIP_C_64F)
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "complex element-wise trinary operations");
    auto same_layout = [&](const TensorUse& X) {
        if (X.modes.size() != op.D.modes.size() || X.desc.dtype != op.D.desc.dtype) return false;
        for (size_t i = 0; i < op.D.modes.size(); ++i) {
            const int ix = find_label(X.modes, op.D.modes[i]);
            if (ix < 0 || X.desc.extent[ix] != op.D.desc.extent[i]) return false;
            if (op.D.desc.extent[i] != 1 && X.desc.stride[ix] != op.D.desc.stride[i]) return false;
        }
        return true;
    };
    plan = EwTrinaryPlan{};
    const bool aSame = same_layout(op.A), bSame = same_layout(op.B);
    cutensorOperationDescriptor last = op;      // A := the permuted operand of the last pass, C := C, D := D
    last.kind = OpKind::ElementwiseBinary;
    last.B = TensorUse{};                        // no second tile operand in these forms
    if (aSame || bSame) {
        plan.twoPass = false;
        plan.swapAB = !aSame;                    // E = B, permuted operand = A
        last.A = plan.swapAB ? op.A : op.B;
        // E rides along with D's strides and is read with D's lane width: its pointer alignment bounds the vector variants too
        const TensorUse& eop = plan.swapAB ? op.B : op.A;
        if (eop.desc.alignment % 16 != 0) last.D.desc.alignment = std::min<uint32_t>(last.D.desc.alignment, eop.desc.alignment);
    } else {
        // both permuted: one pass if the tile decomposition of (B -> D) also reads A with 16-byte lanes (the sample's
        // A_{c,b,a}, B_{c,a,b} -> D_{a,b,c} share the partner mode c), 4 |D| bytes instead of 6
        cutensorOperationDescriptor both = op;
        both.kind = OpKind::ElementwiseBinary;
        both.A = op.A;            // tile path
        both.B = op.B;            // second tile (X)
        EwPlan one;
        if (plan_elementwise(both, one, nullptr, false) == CUTENSOR_STATUS_SUCCESS && one.usesX && one.variant == EW_TRANSPOSE) {
            plan.twoPass = false;
            plan.bothPermuted = true;
            plan.last = one;
            plan.last.p.opAB = (int32_t)op.opAB;
            plan.last.p.opAC = (int32_t)op.opReduce;
            return CUTENSOR_STATUS_SUCCESS;
        }
        plan.twoPass = true;
        cutensorOperationDescriptor first = op;
        first.kind = OpKind::Permutation;
        first.C = TensorUse{};
        cutensorStatus_t st = plan_elementwise(first, plan.first, why, false);
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
        last.A = op.B;
    }
    cutensorStatus_t st = plan_elementwise(last, plan.last, why, false);
    if (st != CUTENSOR_STATUS_SUCCESS) return st;
    plan.last.p.opAB = (int32_t)op.opAB;
    plan.last.p.opAC = (int32_t)op.opReduce;
    return CUTENSOR_STATUS_SUCCESS;
}

cutensorStatus_t plan_reduction(const cutensorOperationDescriptor& op, uint64_t wsLimit, int numCUs,
                                ReducePlan& plan, std::string* why) {
    auto fail = [&](cutensorStatus_t st, const char* msg) {
        if (why) *why = msg;
        return st;
    };
    const TensorUse &A = op.A, &C = op.C, &D = op.D;
    const bool cplx = D.desc.dtype == HIP_C_32F || D.desc.dtype == HIP_C_64F;   // python/einsum.h:326-343 (unary einsum on complex tensors)
    if (dup_labels(A.modes) || dup_labels(D.modes)) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "repeated mode label inside one tensor");
    if (C.modes != D.modes) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "modes of C and D differ");
    if (C.desc.extent != D.desc.extent) return fail(CUTENSOR_STATUS_INVALID_VALUE, "extents of C and D differ");
    if (A.desc.dtype != D.desc.dtype || C.desc.dtype != D.desc.dtype) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "mixed data types");
    auto unary_ok = [](cutensorOperator_t o) { return o == CUTENSOR_OP_IDENTITY || o == CUTENSOR_OP_CONJ; };   // CONJ: a no-op on real data
    if (!unary_ok(A.op) || !unary_ok(C.op)) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "only the identity and conjugation operators are implemented");
    const int rop = (int)op.opReduce;
    if (rop != CUTENSOR_OP_ADD && rop != CUTENSOR_OP_MUL && rop != CUTENSOR_OP_MAX && rop != CUTENSOR_OP_MIN)
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "reduction operator");
    if (cplx && rop != CUTENSOR_OP_ADD && rop != CUTENSOR_OP_MUL) return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "MAX / MIN are not defined on complex data");
    for (size_t i = 0; i < D.modes.size(); ++i) {
        const int ia = find_label(A.modes, D.modes[i]);
        if (ia < 0 && D.desc.extent[i] != 1) return fail(CUTENSOR_STATUS_INVALID_VALUE, "output mode missing from A");
        if (ia >= 0 && A.desc.extent[ia] != D.desc.extent[i]) return fail(CUTENSOR_STATUS_INVALID_VALUE, "extent mismatch");
    }

    std::vector<EwMode> kept, red;
    for (size_t i = 0; i < A.modes.size(); ++i) {
        EwMode m;
        m.extent = A.desc.extent[i];
        m.sA = A.desc.stride[i];
        if (m.extent <= 0) return fail(CUTENSOR_STATUS_INVALID_VALUE, "non-positive extent");
        if (m.extent == 1) continue;
        const int id = find_label(D.modes, A.modes[i]);
        if (id >= 0) {
            m.sD = D.desc.stride[id];
            m.sC = C.desc.stride[id];
            kept.push_back(m);
        } else {
            red.push_back(m);
        }
    }
    plan = ReducePlan{};
    if (red.empty()) {
        // pure permutation (einsum.cu:449-450 routes "nij->ijn" here): D = alpha*perm(A) + beta*C
        cutensorOperationDescriptor e = op;
        e.kind = OpKind::ElementwiseBinary;
        e.opReduce = CUTENSOR_OP_ADD;   // nothing is reduced: opReduce (MAX / MIN / MUL) must not become the A-with-C combiner
        e.C.present = true;
        plan.isPermutation = true;
        return plan_elementwise(e, plan.perm, why);
    }
    auto bySA = [](const EwMode& x, const EwMode& y) { return x.sA < y.sA; };
    std::stable_sort(kept.begin(), kept.end(), bySA);
    std::stable_sort(red.begin(), red.end(), bySA);
    fuse(kept, true);
    {   // reduced modes: only A matters
        std::vector<EwMode> out;
        for (const EwMode& m : red) {
            if (!out.empty() && m.sA == out.back().sA * out.back().extent &&
                out.back().extent * m.extent < (1ll << 31)) {
                out.back().extent *= m.extent;
                continue;
            }
            out.push_back(m);
        }
        red.swap(out);
    }
    ReduceParams& p = plan.p;
    std::memset(&p, 0, sizeof(p));
    if (!fill_rest(p.kept, kept) || !fill_rest(p.red, red))
        return fail(CUTENSOR_STATUS_NOT_SUPPORTED, "mode group too large");
    p.op = rop;
    p.conjA = (cplx && A.op == CUTENSOR_OP_CONJ) ? 1 : 0;
    p.conjC = (cplx && C.op == CUTENSOR_OP_CONJ) ? 1 : 0;

    const bool acc64 = (op.compute != nullptr && op.compute->id == 5 /*64F*/) || A.desc.dtype == HIP_R_64F;
    const bool aligned = A.desc.alignment % 16 == 0;
    // the tiled kernels (reduce.hip): a lane owns 16 bytes = nv elements along A's stride-1 mode — fp32 4; fp64 / complex64 2, complex128 1,
    // bf16 / fp16 8 (round 6: wide_elem.h); a compute type wider than the data's own (fp32 / complex64 data accumulated in 64 bits) keeps
    // the generic kernel
    const int64_t nv = 16 / (int64_t)dtype_size(A.desc.dtype);
    const bool wider = op.compute != nullptr && op.compute->id == 5 && A.desc.dtype != HIP_R_64F && A.desc.dtype != HIP_C_64F;
    const bool tiled = aligned && !wider;
    auto others_mult = [&](const EwMode* except) {
        for (const EwMode& m : kept) if (&m != except && m.sA % nv != 0) return false;
        for (const EwMode& m : red)  if (&m != except && m.sA % nv != 0) return false;
        return true;
    };
    plan.variant = RED_GENERIC;
    if (tiled) {
        if (!kept.empty() && kept[0].sA == 1 && kept[0].extent % nv == 0 && others_mult(&kept[0])) plan.variant = RED_COL;
        else if (red[0].sA == 1 && red[0].extent % nv == 0 && others_mult(&red[0])) plan.variant = RED_ROW;
    }

    // ---- how far to split the reduced range ------------------------------------------------
    const uint64_t keptTot = p.kept.total, redTot = p.red.total;
    uint64_t items, wantItems, minRedPerSplit, gran;
    if (plan.variant == RED_COL)      { items = keptTot / (uint64_t)nv; wantItems = (uint64_t)numCUs * 1024; minRedPerSplit = 32;   gran = 4; }
    else if (plan.variant == RED_ROW) { items = keptTot;     wantItems = (uint64_t)numCUs * 32;   minRedPerSplit = 8192; gran = 256 * (uint64_t)nv; }
    else if (!cplx && !red.empty() && red[0].sA == 1 && redTot >= 64) {
        // the element-gather kernel with A's stride-1 mode reduced ('abc->bc', 'ab->b' at odd extents): a wave per kept element, lanes
        // along that mode (reduce.hip reduce_row_any_kernel; 0.4-0.6 -> 2-3 TB/s, profiles/r06zzs_reduce_odd.jsonl)
        p.rowAny = 1u;
        items = keptTot; wantItems = (uint64_t)numCUs * 32; minRedPerSplit = 4096; gran = 256;
    }
    else                              { items = keptTot;     wantItems = (uint64_t)numCUs * 512;  minRedPerSplit = 64;   gran = 4; }
    uint64_t split = 1;
    if (items < wantItems) split = (wantItems + items - 1) / std::max<uint64_t>(items, 1);
    split = std::min<uint64_t>(split, std::max<uint64_t>(redTot / minRedPerSplit, 1));
    split = std::min<uint64_t>(split, 4096);
    // bytes of one partial: the accumulator type (float / double; (re, im) pairs in the data's precision for complex tensors)
    const uint64_t accBytes = A.desc.dtype == HIP_C_64F ? 16 : A.desc.dtype == HIP_C_32F ? 8 : (acc64 ? 8 : 4);
    while (split > 1 && split * keptTot * accBytes > wsLimit) split /= 2;
    uint64_t per = (redTot + split - 1) / split;
    per = ((per + gran - 1) / gran) * gran;
    split = (redTot + per - 1) / per;
    p.splitR = (uint32_t)std::max<uint64_t>(split, 1);
    p.redPerSplit = (uint32_t)per;
    plan.workspace = (p.splitR > 1) ? (uint64_t)p.splitR * keptTot * accBytes : 0;
    return CUTENSOR_STATUS_SUCCESS;
}

}  // namespace ctamd
