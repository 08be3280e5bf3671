// gett_h16x_common.h — primitives shared by the lean four-wave 16-bit kernels (gett_h16v.hip: gett_h16w4v / w4x / w4m / w8m / w4q;
// gett_h16p.hip: the persistent gett_h16w4p_kernel): the LDS-DMA piece with a literal M0 offset, fragment reads with immediate
// offsets, the K odometer over the descriptor bases, the inline-asm 16x16x32 MFMA and the fragment address maps.
#pragma once
#include "gett_h16_common.h"

namespace ctamd {

typedef __attribute__((address_space(3))) const s16x8* VLdsVec8;
typedef s16x4 __attribute__((address_space(3))) * VLdsVec4;

// 64 lanes x 16 B -> the 1-KiB LDS piece at byte address waveLds + IMM (waveLds wave-uniform).  PAD: the SGPR operands may come
// straight from a v_readfirstlane (VALU-write -> VMEM-read hazard).  Completion is counted by hand (CTAMD_H_VMCNT).
template <uint32_t IMM, bool PAD>
__device__ __forceinline__ void v_dma16(HRsrc rsrc, uint32_t laneBytes, uint32_t waveLds) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PAD)
        asm volatile("s_nop 4\n\ts_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     :: "s"(waveLds), "v"(laneBytes), "s"(rsrc), "i"(IMM) : "memory", "scc");
    else
        asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     :: "s"(waveLds), "v"(laneBytes), "s"(rsrc), "i"(IMM) : "memory", "scc");
#else
    (void)rsrc; (void)laneBytes; (void)waveLds;
#endif
}

// RAG (ragged K, gett_h16w4x_kernel<..., RAG = true>): 2^31 records instead of 2^32 - 1 — a lane whose 16 bytes lie past the end of
// the contracted mode in the last K-tile carries bit 31 in its byte offset, is out of range, touches no memory and has zeros
// written to its 16 bytes of the LDS piece (every in-range offset is below 2^31: pick_h16_choice, tile_span_bytes).
template <bool RAG = false>
__device__ __forceinline__ HRsrc v_rsrc(uint64_t addr) {      // addr is a valid device address: bits 48..63 are zero
    HRsrc r;
    r[0] = (int)(uint32_t)addr;
    r[1] = (int)(uint32_t)(addr >> 32);
    r[2] = RAG ? (int)0x80000000u : -1;
    r[3] = 0x00020000;
    return r;
}

// ... with the record count from the odometer (VOdometer::recs: 0xffffffff, or 0 once the K range is exhausted — every lane out of
// range: no memory access, zeros in LDS.  gett_h16w4p_kernel, round 6: the zero K-tile that pads an odd K-tile count)
__device__ __forceinline__ HRsrc v_rsrc_n(uint64_t addr, uint32_t recs) {
    HRsrc r;
    r[0] = (int)(uint32_t)addr;
    r[1] = (int)(uint32_t)(addr >> 32);
    r[2] = (int)recs;
    r[3] = 0x00020000;
    return r;
}

// One fragment (32 rows x 16 k) from LDS byte address base + IMM.  LAY_K: one ds_read_b128; LAY_F: two transposing reads.
template <int LAY, int IMM>
__device__ __forceinline__ s16x8 v_read(uint32_t base) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LAY == LAY_K) {
        return *(VLdsVec8)(uintptr_t)(base + (uint32_t)IMM);
    } else {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VLdsVec4)(uintptr_t)(base + (uint32_t)IMM));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VLdsVec4)(uintptr_t)(base + (uint32_t)IMM + 1024u));
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#else
    (void)base; return s16x8{};
#endif
}

// K odometer over the descriptor bases (bytes).  K index = j0 * 64 + E0 * (j1 + e1 * hi) as in HOdometer; here digit 0 is a
// countdown and `untilEvent` counts the advances up to the next rare event.  Exactly nTiles - 1 advances move the bases
// (K-tiles 1 .. nTiles - 1 of this workgroup's slice); any further call leaves them where they are (the last tile is re-staged,
// never read).
struct VOdometer {
    uint64_t addrA, addrB, stepA, stepB, wrapA, wrapB;     // hot
    uint32_t untilWrap, n0, untilEvent;                    // hot
    uint32_t left, carryLen, hi, e1, carryPending;         // cold (event path)
    uint32_t recs, ended;                                  // records of the descriptors (v_rsrc_n); the last valid advance has happened
    uint64_t baseA, baseB;                                 // cold: descriptor bases at K index 0

    __device__ static __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

    // RAG: the (single) contracted mode ends inside its last K-tile — that tile counts
    template <bool RAG = false>
    __device__ __forceinline__ void init(const ModeGroup& gK, uint32_t k0, uint32_t nTiles, uint64_t bA, uint64_t bB) {
        const uint32_t E0 = gK.div[0].d;
        n0 = sgpr((E0 + (RAG ? (uint32_t)kHBK - 1u : 0u)) / kHBK);
        e1 = sgpr(gK.div[1].d);
        const uint32_t q0 = (E0 < 2) ? k0 : fast_div(k0, gK.div[0]);
        const uint32_t j0 = (k0 - q0 * E0) / kHBK;
        hi = sgpr((e1 < 2) ? q0 : fast_div(q0, gK.div[1]));
        const uint32_t j1 = q0 - hi * e1;
        baseA = bA;
        baseB = bB;
        addrA = h_uniform64(bA + (uint64_t)(group_offset<0>(gK, k0) * 2));
        addrB = h_uniform64(bB + (uint64_t)(group_offset<1>(gK, k0) * 2));
        stepA = h_uniform64((uint64_t)((int64_t)kHBK * gK.stride[0][0] * 2));
        stepB = h_uniform64((uint64_t)((int64_t)kHBK * gK.stride[1][0] * 2));
        wrapA = h_uniform64((uint64_t)(gK.stride[0][1] * 2) - (uint64_t)(n0 - 1) * stepA);
        wrapB = h_uniform64((uint64_t)(gK.stride[1][1] * 2) - (uint64_t)(n0 - 1) * stepB);
        untilWrap = sgpr(n0 - j0);
        carryLen = sgpr(n0 * e1);
        const uint32_t untilCarry = (e1 - j1) * n0 - j0;
        left = sgpr(nTiles - 1u);
        carryPending = 0;
        untilEvent = 1;
        recs = 0xffffffffu;
        ended = 0;
        next_segment(untilCarry);
    }
    // Sweep-ragged K (round 6, GettParams::ragged bit 1): SEVERAL contracted modes and the fastest one (extent E0) does not hold whole
    // K-tiles — every sweep of that mode is ceil(E0 / 64) K-tiles and its last one is staged masked (x_rag_toggle).  The K-tiles are
    // counted in that padded space: tile T = digit-0 tile T % n0 of the index q0 = T / n0 over the other contracted modes.
    __device__ __forceinline__ void init_tiles(const ModeGroup& gK, uint32_t tile0, uint32_t nTiles, uint64_t bA, uint64_t bB) {
        const uint32_t E0 = gK.div[0].d;
        n0 = sgpr((E0 + (uint32_t)kHBK - 1u) / kHBK);
        e1 = sgpr(gK.div[1].d);
        const uint32_t q0 = (n0 == 1u) ? tile0 : tile0 / n0;
        const uint32_t j0 = tile0 - q0 * n0;
        const uint32_t k0 = q0 * E0 + j0 * (uint32_t)kHBK;
        hi = sgpr((e1 < 2) ? q0 : fast_div(q0, gK.div[1]));
        const uint32_t j1 = q0 - hi * e1;
        baseA = bA;
        baseB = bB;
        addrA = h_uniform64(bA + (uint64_t)(group_offset<0>(gK, k0) * 2));
        addrB = h_uniform64(bB + (uint64_t)(group_offset<1>(gK, k0) * 2));
        stepA = h_uniform64((uint64_t)((int64_t)kHBK * gK.stride[0][0] * 2));
        stepB = h_uniform64((uint64_t)((int64_t)kHBK * gK.stride[1][0] * 2));
        wrapA = h_uniform64((uint64_t)(gK.stride[0][1] * 2) - (uint64_t)(n0 - 1) * stepA);
        wrapB = h_uniform64((uint64_t)(gK.stride[1][1] * 2) - (uint64_t)(n0 - 1) * stepB);
        untilWrap = sgpr(n0 - j0);
        carryLen = sgpr(n0 * e1);
        const uint32_t untilCarry = (e1 - j1) * n0 - j0;
        left = sgpr(nTiles - 1u);
        carryPending = 0;
        untilEvent = 1;
        recs = 0xffffffffu;
        ended = 0;
        next_segment(untilCarry);
    }
    // the tile the bases are on is the last one of a sweep of the fastest contracted mode (between advances; meaningless once the valid
    // advances are used up — the callers stop looking then)
    __device__ __forceinline__ bool on_sweep_end() const { return untilWrap == 1u; }
    // the valid advances that are left are cut into segments that end at a carry past digit 1 or at the end of the K range
    __device__ __forceinline__ void next_segment(uint32_t toCarry) {
        if (left == 0u) {
            // no valid advance is left: the bases stay where they are.  ONE more event, at the next advance (the first one past the range),
            // closes the descriptors (recs = 0): what is staged from then on is zeros and reads no memory — the re-staged, never-read tiles
            // at the end of a K loop, and the zero K-tile the persistent kernel appends to an odd K-tile count
            stepA = stepB = wrapA = wrapB = 0;
            carryPending = 0;
            if (ended == 0u) { ended = 1u; untilEvent = 1u; }
            else { recs = 0u; untilEvent = 0x7fffffffu; }
        } else {
            const uint32_t seg = toCarry < left ? toCarry : left;
            left = sgpr(left - seg);
            carryPending = sgpr(seg == toCarry ? 1u : 0u);
            untilEvent = sgpr(seg);
        }
    }
    __device__ __forceinline__ void advance_a() {          // digit 0 and the A base
        untilWrap -= 1u;
        addrA += (untilWrap == 0u) ? wrapA : stepA;
    }
    __device__ __forceinline__ void advance_b() {          // the B base, digit 0 reloaded
        addrB += (untilWrap == 0u) ? wrapB : stepB;
        untilWrap = (untilWrap == 0u) ? n0 : untilWrap;
    }
    __device__ __forceinline__ void advance_event(const ModeGroup& gK) {
        untilEvent -= 1u;
        if (__builtin_expect(untilEvent == 0u, 0)) {
            if (carryPending != 0u) {                      // the advance just made wrapped digit 1: bases from the full index
                hi += 1u;
                const uint32_t k = hi * e1 * gK.div[0].d;
                if (k < gK.total) {
                    addrA = h_uniform64(baseA + (uint64_t)(group_offset<0>(gK, k) * 2));
                    addrB = h_uniform64(baseB + (uint64_t)(group_offset<1>(gK, k) * 2));
                }
            }
            next_segment(carryLen);
        }
    }
};

// Ragged K / operands without 16-byte lanes (the RAG instantiations of gett_h16w4x_kernel / gett_h16w4m_kernel; pick_h16_choice).  The
// LAST K-tile of the last slice is staged with the lanes that must not read memory out of range — bit 31 or-ed into their byte offsets,
// descriptors of 2^31 records (v_rsrc<true>): no memory access, zeros in their 16 bytes of the LDS piece.  Which lanes:
//   * k past the end of the contracted range (round 5).  K-contiguous operand: a lane's 16 bytes are k-unit
//     u = (lane & 7) ^ ((4 wave + (lane >> 4)) & 7) of its row in every piece (HOperand::init: r >> 1 = 4 wave + 16 i + (lane >> 4)) — out
//     when 8 u >= kValid; free-contiguous operand: the lanes of piece i hold k-row 16 i + 4 wave + (lane >> 4) — out when that is >= kValid;
//   * round 6: a unit that holds live data but whose 16 bytes reach past the END OF THE TENSOR (`limit` = bytes from the descriptor base of
//     that tile to the end, wave-uniform, clamped to 2^31 - 1).  Possible since operands need no 16-byte lanes any more: the partial last
//     k-unit of the last row (K-contiguous, K % 8 != 0) and the partial last row-unit of the last k-row (free-contiguous, extent % 8 != 0).
//     These lanes are returned as a bit set (bit 4 h + i) and x_rag_fix loads their live elements one by one.
// A partial k-unit that ends INSIDE the tensor (every other row of a K-contiguous operand with K % 8 != 0) is staged whole — its tail is the
// head of the next row — and x_rag_fix zeroes the tail in LDS once the tile has landed.  Called ONCE per operand, right before the first
// LDS-DMA piece of that tile, with the descriptor base already on that tile: nothing is staged after the last tile but itself.
template <int LAY, int NH>
__device__ __forceinline__ uint32_t x_rag_mask(uint32_t (&src)[NH][4], int wave, uint32_t kValid, uint32_t limit) {
    const uint32_t laneM = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t kk0 = 4u * (uint32_t)wave + (laneM >> 4);
    uint32_t strad = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool out = (LAY == LAY_K) ? (8u * ((laneM & 7u) ^ (kk0 & 7u)) >= kValid) : (kk0 + 16u * (uint32_t)i >= kValid);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const bool past = !out && src[h][i] + 16u > limit;
            strad |= past ? (1u << (4 * h + i)) : 0u;
            src[h][i] |= (out || past) ? 0x80000000u : 0u;
        }
    }
    return strad;
}
// Sweep-ragged K: the mask of the lanes past the end of the fastest contracted mode, switched ON for the last K-tile of every sweep and OFF
// again for the next sweep's first tile (bit 31 of the lane offsets flipped: x_rag_mask's `out` lanes — the planner admits this form only
// when no 16-byte unit is partial or can reach past the end of the tensor, so there is no `past` lane and nothing for x_rag_fix to do).
template <int LAY, int NH>
__device__ __forceinline__ void x_rag_toggle(uint32_t (&src)[NH][4], int wave, uint32_t kValid) {
    const uint32_t laneM = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t kk0 = 4u * (uint32_t)wave + (laneM >> 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool out = (LAY == LAY_K) ? (8u * ((laneM & 7u) ^ (kk0 & 7u)) >= kValid) : (kk0 + 16u * (uint32_t)i >= kValid);
#pragma unroll
        for (int h = 0; h < NH; ++h) src[h][i] ^= out ? 0x80000000u : 0u;
    }
}
// bytes from `addr` (the descriptor base of the masked tile) to `end`, clamped to what a 31-bit lane offset can reach
__device__ __forceinline__ uint32_t x_rag_limit(uint64_t end, uint64_t addr) {
    const uint64_t d = end - addr;
    return VOdometer::sgpr(d > 0x7fffffffull ? 0x7fffffffu : (uint32_t)d);
}

typedef uint16_t __attribute__((address_space(3))) * XLdsU16;
typedef uint32_t __attribute__((address_space(3))) * XLdsU32;
// zero elements v .. 7 of the 16-byte unit at LDS byte address `at` (0 < v < 8): one 2-byte write when v is odd, then whole dwords
__device__ __forceinline__ void x_zero_tail(uint32_t at, uint32_t v) {
    if (v & 1u) *(XLdsU16)(uintptr_t)(at + 2u * v) = (uint16_t)0;
    const uint32_t w = (v + 1u) >> 1;
    if (w <= 1u) *(XLdsU32)(uintptr_t)(at + 4u) = 0u;
    if (w <= 2u) *(XLdsU32)(uintptr_t)(at + 8u) = 0u;
    if (w <= 3u) *(XLdsU32)(uintptr_t)(at + 12u) = 0u;
}
// elements 0 .. nv - 1 of the unit at byte address `addr` into the (zero-filled) unit at LDS byte address `at`
__device__ __forceinline__ void x_patch_unit(uint32_t at, uint64_t addr, uint32_t nv) {
    const HGlbCU16 g = (HGlbCU16)(uintptr_t)addr;
#pragma unroll 1
    for (uint32_t e = 0; e < nv; ++e) *(XLdsU16)(uintptr_t)(at + 2u * e) = g[e];
}
// The masked tile has landed (every wave's pieces, behind a workgroup barrier): repair what whole 16-byte units could not express.
//   K-contiguous operand: the tail of a partial k-unit (elements kValid - 8 u .. 7) is zeroed; a partial unit x_rag_mask kept from memory
//   (strad) gets its live elements by 2-byte loads.   Free-contiguous operand: only strad units (live rows of the mode, by 2-byte loads).
// opBase: byte address of the operand's element (batch index l, row 0, k 0); row0: first row of the workgroup's tile; kTile0: K index of
// the tile's first k; ldsOp: LDS byte address of the operand's half-tile 0 in the buffer that holds the tile.  The caller follows up with
// s_waitcnt + a workgroup barrier.  SLOTK: the operand's stride slot in the K group (0 = kernel-A, 1 = kernel-B).
template <int LAY, int NH, int SLOTK>
__device__ __forceinline__ void x_rag_fix(const ModeGroup& gFree, const ModeGroup& gK, uint64_t opBase, uint32_t row0, uint32_t kTile0, uint32_t kValid,
                                          uint32_t strad, uint32_t ldsOp, int wave) {
    const uint32_t laneM = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t kk0 = 4u * (uint32_t)wave + (laneM >> 4);
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = (uint32_t)wave + 4u * (uint32_t)i;               // 1-KiB piece of the half-tile
            const uint32_t at = ldsOp + (uint32_t)h * (uint32_t)kHalfBytes + c * 1024u + laneM * 16u;
            const bool past = ((strad >> (4 * h + i)) & 1u) != 0u;
            if constexpr (LAY == LAY_K) {
                const uint32_t u = (laneM & 7u) ^ (kk0 & 7u);
                const uint32_t v = kValid > 8u * u ? (kValid - 8u * u < 8u ? kValid - 8u * u : 8u) : 0u;
                if (!past && v > 0u && v < 8u) x_zero_tail(at, v);
                if (past) {
                    uint32_t row = row0 + 128u * (uint32_t)h + 8u * c + (laneM >> 3);
                    if (row >= gFree.total) row = gFree.total - 1u;
                    const uint64_t addr = opBase + (uint64_t)((group_offset<0>(gFree, row) + group_offset<SLOTK>(gK, kTile0 + 8u * u)) * 2);
                    x_patch_unit(at, addr, v);
                }
            } else {
                if (past) {
                    const uint32_t kk = kk0 + 16u * (uint32_t)i;
                    const uint32_t u = (laneM & 15u) ^ (4u * ((laneM >> 4) & 3u)) ^ (2u * ((kk >> 3) & 1u));
                    uint32_t row = row0 + 128u * (uint32_t)h + 8u * u;
                    if (row >= gFree.total) row = (gFree.total - 1u) & ~7u;
                    const uint32_t nv = gFree.total - row < 8u ? gFree.total - row : 8u;
                    const uint64_t addr = opBase + (uint64_t)((group_offset<0>(gFree, row) + group_offset<SLOTK>(gK, kTile0 + kk)) * 2);
                    x_patch_unit(at, addr, nv);
                }
            }
        }
}

template <bool BF>
__device__ __forceinline__ void x_mfma(f32x4& c, const s16x8& a, const s16x8& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (BF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#else
    (void)c; (void)a; (void)b;
#endif
}
// The accumulators as the epilogue may read them: every fragment passes through an (empty) asm statement placed BEHIND the two
// s_nop 15 that wait out the last MFMAs — volatile asm statements keep their order, and every later read of an accumulator depends
// on this one.  Without it nothing ties the epilogue's v_accvgpr_read copies to the s_nops: the compiler has hoisted them in front
// of the wait (tests/test_kernel_resources.py checks the order in every instantiation).
template <int FI, int FJ>
__device__ __forceinline__ void x_acc_ready(f32x4 (&acc)[FI][FJ]) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs have written their accumulators
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) asm volatile("" : "+a"(acc[i][j]));
#endif
}

// byte offset of this lane's 16 bytes of a 16-row fragment, k-step s (0, 1), inside a K-contiguous half-tile (rows 16 f: + 2048 f)
__device__ __forceinline__ uint32_t x_offK(int lane, int s) {
    const int row = lane & 15, unit = (lane >> 4) + 4 * s;
    return (uint32_t)(row * 128 + (((unit ^ (row >> 1)) & 7) << 4));
}

// the same for a free-contiguous half-tile (image [64 k][128 rows], 256-byte k-rows, HOperand's SWZ = 1 form: unit p of k-row k holds
// row-unit p ^ 4 (k & 3) ^ 2 ((k >> 3) & 1)): a 16-lane group g fetches the 4 k x 16 rows block (k = 8 g + [0,4), rows of fragment f)
// that ds_read_b64_tr_b16 turns into "lane = row, four k"; the second read (+ 1024 B) brings k + 4.  The block is 32 contiguous
// bytes in each of its four k-rows: quarter (f >> 1) ^ (k & 3) of the row, half (f & 1) ^ (g & 1) of the quarter — the groups g and
// g + 1 that one read serves together never share a bank.  Both f >> 1 and f & 1 sit inside the swizzle: one address register per
// (f >> 1, f & 1), k-step and + 4 as immediates (8192 s, 1024).
__device__ __forceinline__ uint32_t x_offF(int lane, int f) {
    const int g = lane >> 4, i = lane & 15, q = (i >> 2) & 3, b = (i >> 1) & 1;
    const int unit = b | ((((f & 1) ^ (g & 1)) & 1) << 1) | ((((f >> 1) ^ q) & 3) << 2);
    return (uint32_t)((8 * g + (i >> 2)) * 256 + (unit << 4) + 8 * (i & 1));
}

}  // namespace ctamd
