// gett_h16v.hip — the four-wave 16-bit GETT kernel with a LEAN instruction stream (CUTENSOR_AMD_H16_WAVES=4v; round 3).
//
// Same arithmetic, tile (256 x 256 x 64), LDS images, source-side swizzles, MFMA order, barrier placement and epilogue as
// gett_h16w4_kernel (gett_h16.hip): 4 waves as 2 (M) x 2 (N), one per SIMD, a 128 x 128 quadrant each (4 x 4 accumulator
// fragments, 256 AGPRs), two 64-deep K-tiles in LDS, ONE barrier per K-tile.  What changes is everything BETWEEN the MFMAs.
// With one wave per SIMD every instruction the wave issues sits in front of its own MFMAs (nothing else fills the issue slots),
// and the matrix pipe only stays busy while at most ~6 other instructions separate two MFMAs.  The compiler's version of the
// four-wave loop carried ~230 non-MFMA instructions per K-tile (64 MFMAs), the vendor's hand-written kernel of the same
// structure ~95 (NOTES.md, profiles/r03_h16_vendor_pmc_instruction_mix.txt).  Here:
//   * LDS-DMA destination: M0 = (one SGPR: ring base + wave * 1 KiB) + a literal, formed by the s_add_u32 that writes M0 —
//     instead of 32 loop-invariant SGPR addresses, most of which lived in VGPR lanes (v_readlane -> s_mov m0 per piece);
//   * fragment reads: 16 address registers (operand x buffer x k-step or fragment), every other term in the 16-bit offset
//     field of the ds_read — instead of a v_add per read;
//   * K odometer: the descriptor bases themselves advance (s_add_u32 / s_addc_u32 by a selected step), digit 0 by a countdown,
//     and ONE countdown covers both rare events (carry past the second K digit, end of the K range: steps become zero);
//     ~14 scalar instructions per K-tile spread over three MFMA pairs — instead of ~60 in one block behind a branch.
// Roofline and algorithmic bytes as in gett_h16.hip.
#include <type_traits>

#include "gett_h16x_common.h"

namespace ctamd {

#if defined(CTAMD_RESEARCH_KERNELS)   // gett_h16w4v_kernel: the 32x32x16 sibling of the default, retired in round 5 (research builds only)
template <bool BF, int LA, int LB>
__global__ void __launch_bounds__(256, 1) gett_h16w4v_kernel(const GettParams p) {
    __shared__ __attribute__((aligned(16))) char lds[8 * kHalfBytes];
    prefetch_kernarg<(int)sizeof(GettParams)>();
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    uint32_t id = xcd_remap(blockIdx.x, p.nBlocks);
    const uint32_t tilesMN = p.tilesM * p.tilesN;
    const uint32_t tilesAll = tilesMN * p.gL.total;
    const uint32_t slice = id / tilesAll;
    id -= slice * tilesAll;
    const uint32_t l = id / tilesMN;
    id -= l * tilesMN;
    const uint32_t perGroup = 8u * p.tilesN;
    const uint32_t grp = id / perGroup, inGrp = id - grp * perGroup;
    const uint32_t first = grp * 8u;
    const uint32_t gsz = (p.tilesM - first < 8u) ? (p.tilesM - first) : 8u;
    const uint32_t mt = first + inGrp % gsz, nt = inGrp / gsz;
    const uint32_t m0 = mt * kHTile, n0 = nt * kHTile;
    const uint32_t kTilesAll = p.gK.total / kHBK, tilesPerSlice = p.kPerSlice / kHBK;
    const uint32_t tile0 = slice * tilesPerSlice;
    const int nTiles = (int)((tile0 + tilesPerSlice <= kTilesAll) ? tilesPerSlice : (kTilesAll - tile0));

    HOperand<LA, 4> oa;
    HOperand<LB, 4> ob;
    oa.init(p.gM, p.gK.stride[0][0], m0, wave, lane);
    ob.init(p.gN, p.gK.stride[1][0], n0, wave, lane);
    // descriptor base = operand + batch offset + this wave's smallest piece offset (+ the K-tile's offset: the odometer)
    const uint64_t bA = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)) + oa.base);
    const uint64_t bB = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)) + ob.base);
    VOdometer odo;
    odo.init(p.gK, tile0 * kHBK, (uint32_t)nTiles, bA, bB);

    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t waveLds = VOdometer::sgpr(ldsBase + (uint32_t)wave * 1024u);

    // fragment-read address registers: [buffer][k-step] for a K-contiguous operand (immediate: 4096 x fragment), [buffer][fragment]
    // for a free-contiguous one (immediate: 4096 x k-step); opaque, so that they stay registers instead of becoming an add per read
    uint32_t rdA[2][4], rdB[2][4];
#pragma unroll
    for (int P = 0; P < 2; ++P)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            rdA[P][x] = ldsBase + (uint32_t)((P * 4 + wr) * kHalfBytes) + (LA == LAY_K ? h_offK(lane, x) : h_offF(lane, x));
            rdB[P][x] = ldsBase + (uint32_t)((P * 4 + 2 + wc) * kHalfBytes) + (LB == LAY_K ? h_offK(lane, x) : h_offF(lane, x));
            asm volatile("" : "+v"(rdA[P][x]));
            asm volatile("" : "+v"(rdB[P][x]));
        }

    // piece N = 0..15 of the K-tile the odometer describes, into buffer P: operand half q = N >> 2 (A0, A1, B0, B1), piece i = N & 3
    // of this wave (1-KiB piece wave + 4 i of the half-tile)
#define CTAMD_V_DMA(P, N, PAD)                                                                                      \
    {                                                                                                              \
        constexpr int q_ = (N) >> 2, i_ = (N) & 3;                                                                 \
        constexpr uint32_t imm_ = (uint32_t)(((P) * 4 + q_) * kHalfBytes + i_ * 4096);                             \
        if constexpr (q_ < 2) v_dma16<imm_, PAD>(v_rsrc(odo.addrA), oa.src[q_][i_], waveLds);                      \
        else v_dma16<imm_, PAD>(v_rsrc(odo.addrB), ob.src[q_ - 2][i_], waveLds);                                   \
    }
#define CTAMD_V_DMA8(P, N0, PAD)                                                                                    \
    CTAMD_V_DMA(P, (N0) + 0, PAD) CTAMD_V_DMA(P, (N0) + 1, PAD) CTAMD_V_DMA(P, (N0) + 2, PAD) CTAMD_V_DMA(P, (N0) + 3, PAD) \
    CTAMD_V_DMA(P, (N0) + 4, PAD) CTAMD_V_DMA(P, (N0) + 5, PAD) CTAMD_V_DMA(P, (N0) + 6, PAD) CTAMD_V_DMA(P, (N0) + 7, PAD)
#define CTAMD_V_ADVANCE() { odo.advance_a(); odo.advance_b(); odo.advance_event(p.gK); }

    // ---- prologue: K-tile 0 complete, the first half of K-tile 1 ---------------------------------------------
    CTAMD_V_DMA8(0, 0, true) CTAMD_V_DMA8(0, 8, true)
    CTAMD_V_ADVANCE()
    CTAMD_V_DMA8(1, 0, true)
    CTAMD_H_VMCNT(8);                             // this wave's pieces of tile 0
    __builtin_amdgcn_s_barrier();

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    s16x8 a[2][4], b[2][4];                       // two register sets: k-step s uses set s & 1

    // fragment F = 0..7 of k-step S from buffer P into register set SET: F < 4 -> B columns 32 F, else A rows 32 (F - 4)
#define CTAMD_V_READ(P, S, SET, F)                                                                                  \
    {                                                                                                              \
        if constexpr ((F) < 4) {                                                                                   \
            if constexpr (LB == LAY_K) b[SET][F] = v_read<LB, 4096 * (F)>(rdB[P][S]);                              \
            else b[SET][F] = v_read<LB, 4096 * (S)>(rdB[P][F]);                                                    \
        } else {                                                                                                   \
            if constexpr (LA == LAY_K) a[SET][(F) - 4] = v_read<LA, 4096 * ((F) - 4)>(rdA[P][S]);                  \
            else a[SET][(F) - 4] = v_read<LA, 4096 * (S)>(rdA[P][(F) - 4]);                                        \
        }                                                                                                          \
    }
#define CTAMD_V_MFMA(SET, M) acc[(M) >> 2][(M) & 3] = h_mfma<BF>(a[SET][(M) >> 2], b[SET][(M) & 3], acc[(M) >> 2][(M) & 3]);
    // k-step S < 3: one fragment read of step S + 1 per two MFMAs of step S.  k-step 0 also carries the second half (pieces
    // 8..15) of the tile being staged into the other buffer; k-step 1 the odometer (three pairs)
#define CTAMD_V_PAIR(P, S, F)                                                                                       \
    CTAMD_V_READ(P, (S) + 1, ((S) + 1) & 1, F) CTAMD_V_MFMA((S) & 1, 2 * (F))                                      \
    if constexpr ((S) == 0) CTAMD_V_DMA((P) ^ 1, 8 + (F), false)                                                   \
    if constexpr ((S) == 1 && (F) == 1) odo.advance_a();                                                           \
    if constexpr ((S) == 1 && (F) == 3) odo.advance_b();                                                           \
    if constexpr ((S) == 1 && (F) == 5) odo.advance_event(p.gK);                                                   \
    CTAMD_V_MFMA((S) & 1, 2 * (F) + 1)                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#define CTAMD_V_STEP(P, S)                                                                                          \
    CTAMD_V_PAIR(P, S, 0) CTAMD_V_PAIR(P, S, 1) CTAMD_V_PAIR(P, S, 2) CTAMD_V_PAIR(P, S, 3)                        \
    CTAMD_V_PAIR(P, S, 4) CTAMD_V_PAIR(P, S, 5) CTAMD_V_PAIR(P, S, 6) CTAMD_V_PAIR(P, S, 7)
    // k-step 3 (behind the barrier): reads of the next tile's step 0 (other buffer), the first half (pieces 0..7) of tile
    // t + 2 into this buffer, MFMAs of step 3 — a read and a piece alternate, one per MFMA
#define CTAMD_V_LAST2(P, F)                                                                                         \
    CTAMD_V_READ((P) ^ 1, 0, 0, F)                                                                                 \
    CTAMD_V_MFMA(1, 2 * (F)) __builtin_amdgcn_sched_barrier(0);                                                    \
    CTAMD_V_DMA(P, F, false)                                                                                       \
    CTAMD_V_MFMA(1, 2 * (F) + 1) __builtin_amdgcn_sched_barrier(0);
#define CTAMD_V_TILE(P)                                                                                             \
    CTAMD_V_STEP(P, 0) CTAMD_V_STEP(P, 1) CTAMD_V_STEP(P, 2)                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_H_LGKM0();                                                                                               \
    CTAMD_H_VMCNT(0);                                                                                              \
    __builtin_amdgcn_s_barrier();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_V_LAST2(P, 0) CTAMD_V_LAST2(P, 1) CTAMD_V_LAST2(P, 2) CTAMD_V_LAST2(P, 3)                                \
    CTAMD_V_LAST2(P, 4) CTAMD_V_LAST2(P, 5) CTAMD_V_LAST2(P, 6) CTAMD_V_LAST2(P, 7)

    // first fragments of tile 0
    CTAMD_V_READ(0, 0, 0, 0) CTAMD_V_READ(0, 0, 0, 1) CTAMD_V_READ(0, 0, 0, 2) CTAMD_V_READ(0, 0, 0, 3)
    CTAMD_V_READ(0, 0, 0, 4) CTAMD_V_READ(0, 0, 0, 5) CTAMD_V_READ(0, 0, 0, 6) CTAMD_V_READ(0, 0, 0, 7)
    int t = 0;
    for (; t + 1 < nTiles; t += 2) { CTAMD_V_TILE(0) CTAMD_V_TILE(1) }
    if (t < nTiles) { CTAMD_V_TILE(0) }
    CTAMD_H_VMCNT(0);                             // the re-staged tail: no LDS-DMA may outlive the workgroup

    const int laneE = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // fresh: nothing lane-derived lives across the loop for the epilogue
    const uint32_t mW = m0 + 128 * wr, nW = n0 + 128 * wc;    // this wave's quadrant
    if (p.partial != nullptr) {                   // split-K: fp32 partial tile, row-major [slice][l][m][n]
        const uint32_t Mt = p.gM.total, Nt = p.gN.total;
        float* P = p.partial + ((size_t)slice * p.gL.total + l) * (size_t)Mt * Nt;
        auto store_partial = [&](const f32x16& c0, const f32x16& c1, const f32x16& c2, const f32x16& c3, uint32_t mBase) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = mBase + (r & 3) + 8 * (r >> 2) + 4 * (laneE >> 5);
                if (m < Mt) {
                    const uint32_t n = nW + (laneE & 31);
                    float* row = P + (size_t)m * Nt;
                    if (n < Nt) row[n] = c0[r];
                    if (n + 32 < Nt) row[n + 32] = c1[r];
                    if (n + 64 < Nt) row[n + 64] = c2[r];
                    if (n + 96 < Nt) row[n + 96] = c3[r];
                }
            }
        };
        store_partial(acc[0][0], acc[0][1], acc[0][2], acc[0][3], mW);
        store_partial(acc[1][0], acc[1][1], acc[1][2], acc[1][3], mW + 32);
        store_partial(acc[2][0], acc[2][1], acc[2][2], acc[2][3], mW + 64);
        store_partial(acc[3][0], acc[3][1], acc[3][2], acc[3][3], mW + 96);
        return;
    }
    __syncthreads();                              // every wave has finished reading the operand ring
    HEpilogue ep;
    ep.init(p, l, lds, wave);
#pragma unroll
    for (int i = 0; i < 4; ++i) {                 // four passes: the four fragments of accumulator row i
        ep.park(0, acc[i][0], laneE); ep.park(1, acc[i][1], laneE); ep.park(2, acc[i][2], laneE); ep.park(3, acc[i][3], laneE);
        const uint32_t mB = mW + 32 * i;
        ep.template flush<BF>(p, mB, 0u, 0u, nW, 64u, 32u, laneE);
    }
}

#endif  // CTAMD_RESEARCH_KERNELS

// =====================================================================================================
// gett_h16w4x_kernel (CUTENSOR_AMD_H16_WAVES=4x): the kernel above on v_mfma_f32_16x16x32_{bf16,f16}.
// Why the other shape: under the power limit a stream of nothing but 16x16x32 MFMAs sustains 2.03-2.05 PFLOP/s on U(-1,1)
// operands where 32x32x16 sustains 1.78-1.81 (tools/ubench/mfma16_issue.hip) — half the accumulator read-modify-write traffic
// per flop (K = 32 per instruction).  The instruction is issued from inline asm: through the builtin the compiler's hazard
// recognizer spaces independent 4-pass MFMAs 27 cycles apart instead of 16 (same microbenchmark), so the hazards are kept by
// construction here — an accumulator is touched once per 64 MFMAs, an operand register set is rewritten a k-step after its
// last use, and the epilogue waits out the last write.
// Wave tile 128 x 128 = 8 x 8 accumulator fragments of 16 x 16 (256 AGPRs); K-tile 64 = two k-steps of 32; per k-step 64 MFMAs,
// 16 fragment reads (one per four MFMAs) into the other of two register sets (2 x 16 x 4 VGPRs).  Tile t: k-step 0 (reads of
// k-step 1, the odometer), the tile barrier, k-step 1 (reads of tile t + 1's k-step 0 and the 16 LDS-DMA pieces of tile t + 2,
// one per four MFMAs).  Same LDS images as every 16-bit kernel: a 16-row fragment of the K-contiguous image is rows x 4 units
// (lane & 15 = row, lane >> 4 = k-unit, one ds_read_b128); of the free-contiguous image two transposing reads (x_offF).
// =====================================================================================================
// The 16-bit epilogue of a quadrant that lies inside D with one M and one N mode (wave-uniform test by the caller), as a software
// pipeline over its passes of 32 rows x 16 FJ columns: the 16-byte chunks of pass i - 1 wait in registers and are stored one per two
// accumulator fragments of pass i on their way into the image (pitch kPitch 16-bit elements), addresses one addition apart, no bounds
// test.  gett_h16w4x_kernel carries the same loop inline (FJ = 8); measured there: 16.6k -> 12.0-13.7k cycles.
template <bool BF, int FI, int FJ, int kPitch>
__device__ __forceinline__ void x_store_interior16(f32x4 (&acc)[FI][FJ], uint16_t* stage, float alpha, uint16_t* dst, int64_t stepElems, int laneE) {
    constexpr int kChunks = 2 * FJ, kRowsPerIt = 64 / kChunks, NP = FI / 2;   // chunks per image row, rows a wave-read covers, passes
    s16x8 v[2][FJ];
#pragma unroll
    for (int i = 0; i <= NP; ++i) {
#pragma unroll
        for (int g = 0; g < FJ; ++g) {
            if (i < NP) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int f = 2 * g + jj, a2 = f / FJ, j = f % FJ;
                    const f32x4& c = acc[2 * (i < NP ? i : 0) + a2][j];
                    uint16_t* st = stage + (16 * a2 + 4 * (laneE >> 4)) * kPitch + 16 * j + (laneE & 15);
                    st[0] = h_round16<BF>(alpha * c[0]); st[kPitch] = h_round16<BF>(alpha * c[1]);
                    st[2 * kPitch] = h_round16<BF>(alpha * c[2]); st[3 * kPitch] = h_round16<BF>(alpha * c[3]);
                }
            }
            if (i > 0) {
                __builtin_nontemporal_store(v[(i - 1) & 1][g], reinterpret_cast<s16x8*>(dst));
                dst += stepElems;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (i < NP) {
#pragma unroll
            for (int it = 0; it < FJ; ++it)
                v[i & 1][it] = *reinterpret_cast<const s16x8*>(stage + (kRowsPerIt * it + laneE / kChunks) * kPitch + 8 * (laneE % kChunks));
        }
    }
}

// TIMED (measurement-only instantiation, CUTENSOR_AMD_H16_TIMED=1 with the planner's default kernel, layout mk,kn): wave 0 of every
// workgroup records shader cycles at entry / first MFMA / end of the main loop / exit and the wall clock at entry / exit into
// p.timing (the layout tools/h16_wg_timeline.py reads).  XST (measurement, CUTENSOR_AMD_H16_XST, with TIMED only): 0 the default,
// 1 plain instead of nontemporal stores in the epilogue, 3 always the fp32 LDS image (the general path) instead of the 16-bit one,
// 4 the fragment reads of a k-step issued in its first eight groups (main loop 2280 -> 2415 cycles per K-tile), 5 first-round
// workgroups start staggered (epilogue 16.4k -> 15.8k cycles, rate unchanged) — profiles/r04n_w4x_variants.jsonl.
// XST = 8 / 9 / 10 (round 5, ZERO-FILLED operands only: the results are wrong on any other data — the tile barrier loses its
// vmcnt(0) / its s_barrier / both): what of the 2280 - 2048 cycles per K-tile is data latency and what is wave skew.
// RAG: ragged K — the last K-tile of the last slice staged with the lanes past the end of the contracted mode out of range
// (x_rag_mask, gett_h16x_common.h); the offsets are switched right before the first LDS-DMA of that tile.
template <bool BF, int LA, int LB, bool TIMED = false, int XST = 0, bool RAG = false>
__global__ void __launch_bounds__(256, 1) gett_h16w4x_kernel(const GettParams p) {
    __shared__ __attribute__((aligned(16))) char lds[8 * kHalfBytes];
    unsigned long long wgStamp[6] = {0, 0, 0, 0, 0, 0};
    constexpr bool kEarly = (XST == 4);          // measurement: the 16 fragment reads of a k-step in its first 8 groups
    if constexpr (XST == 5) {                    // measurement: first-round workgroups start up to 15 x 1024 cycles apart
        if (blockIdx.x < 256u) {
            const uint32_t steps = (blockIdx.x * 37u) & 15u;
            for (uint32_t i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(16);
        }
    }
    if constexpr (TIMED) { wgStamp[0] = __builtin_readcyclecounter(); wgStamp[4] = wall_clock64(); }
    prefetch_kernarg<(int)sizeof(GettParams)>();
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    uint32_t id = xcd_remap(blockIdx.x, p.nBlocks);
    const uint32_t tilesMN = p.tilesM * p.tilesN;
    const uint32_t tilesAll = tilesMN * p.gL.total;
    const uint32_t slice = id / tilesAll;
    id -= slice * tilesAll;
    const uint32_t l = id / tilesMN;
    id -= l * tilesMN;
    const uint32_t perGroup = 8u * p.tilesN;
    const uint32_t grp = id / perGroup, inGrp = id - grp * perGroup;
    const uint32_t first = grp * 8u;
    const uint32_t gsz = (p.tilesM - first < 8u) ? (p.tilesM - first) : 8u;
    const uint32_t mt = first + inGrp % gsz, nt = inGrp / gsz;
    const uint32_t m0 = p.mOrg + mt * kHTile, n0 = p.nOrg + nt * kHTile;
    // sweep-ragged K (GettParams::ragged bit 1; the padded K-tile count in bits 2..31): several contracted modes, the fastest one without
    // whole K-tiles — the last K-tile of EVERY sweep of that mode is staged masked (VOdometer::init_tiles, x_rag_toggle)
    const bool sweep = RAG && (p.ragged & 2u) != 0u;
    const uint32_t kTilesAll = sweep ? (p.ragged >> 2) : (p.gK.total + (RAG ? (uint32_t)kHBK - 1u : 0u)) / kHBK, tilesPerSlice = p.kPerSlice / kHBK;
    const uint32_t tile0 = slice * tilesPerSlice;
    const int nTiles = (int)((tile0 + tilesPerSlice <= kTilesAll) ? tilesPerSlice : (kTilesAll - tile0));

    HOperand<LA, 4, false, 1> oa;
    HOperand<LB, 4, false, 1> ob;
    oa.init(p.gM, p.gK.stride[0][0], m0, wave, lane);
    ob.init(p.gN, p.gK.stride[1][0], n0, wave, lane);
    const uint64_t bA = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)) + oa.base);
    const uint64_t bB = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)) + ob.base);
    VOdometer odo;
    if (sweep) odo.init_tiles(p.gK, tile0, (uint32_t)nTiles, bA, bB);
    else odo.template init<RAG>(p.gK, tile0 * kHBK, (uint32_t)nTiles, bA, bB);
    // ragged K / operands without 16-byte lanes: index (among this workgroup's K-tiles) of the tile that is staged masked — the last K-tile
    // of the last slice — and how many k of the contracted range it holds (x_rag_mask, x_rag_fix: gett_h16x_common.h)
    const int maskAt = (RAG && !sweep && tile0 + (uint32_t)nTiles == kTilesAll) ? nTiles - 1 : 0x7fffffff;
    const uint32_t kValid = VOdometer::sgpr(sweep ? p.gK.div[0].d - (odo.n0 - 1u) * (uint32_t)kHBK
                                                  : ((p.gK.total % kHBK) != 0u ? p.gK.total % kHBK : (uint32_t)kHBK));
    uint32_t stradA = 0u, stradB = 0u;            // units of the masked tile that x_rag_fix loads element by element
    bool maskOn = false;                          // sweep-ragged K: the lanes past the end of the fastest contracted mode are out of range right now
    // called with the descriptor bases ON tile IDX (sweep: tiles past the slice's last one are re-staged copies of it — the mask stays as it is)
#define CTAMD_X_RAGMASK(IDX)                                                                                        \
    if constexpr (RAG) {                                                                                           \
        if ((IDX) == maskAt) {                                                                                     \
            stradA = x_rag_mask<LA, 2>(oa.src, wave, kValid, x_rag_limit(p.endA, odo.addrA));                      \
            stradB = x_rag_mask<LB, 2>(ob.src, wave, kValid, x_rag_limit(p.endB, odo.addrB));                      \
        }                                                                                                          \
        if (sweep && (IDX) < nTiles && odo.on_sweep_end() != maskOn) {                                             \
            x_rag_toggle<LA, 2>(oa.src, wave, kValid);                                                             \
            x_rag_toggle<LB, 2>(ob.src, wave, kValid);                                                             \
            maskOn = !maskOn;                                                                                      \
        }                                                                                                          \
    }
    // tile IDX (the masked one) has landed in buffer PB, behind a workgroup barrier: repair it when it holds a partial k-unit
    // (K-contiguous operand, K % 8 != 0: every workgroup) or may hold a unit that was kept from memory (free-contiguous operand with a
    // ragged extent: the workgroups on the last rows) — both tests are uniform over the workgroup, as the second barrier demands
#define CTAMD_X_RAGFIX(IDX, PB)                                                                                     \
    if constexpr (RAG) {                                                                                           \
        if ((IDX) == maskAt) {                                                                                     \
            const bool fixA = (LA == LAY_K) ? (kValid & 7u) != 0u : ((p.gM.total & 7u) != 0u && m0 + (uint32_t)kHTile >= p.gM.total);   \
            const bool fixB = (LB == LAY_K) ? (kValid & 7u) != 0u : ((p.gN.total & 7u) != 0u && n0 + (uint32_t)kHTile >= p.gN.total);   \
            if (fixA || fixB) {                                                                                    \
                const uint32_t kT0 = (kTilesAll - 1u) * (uint32_t)kHBK;                                            \
                if (fixA) x_rag_fix<LA, 2, 0>(p.gM, p.gK, (uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)), m0, kT0, kValid, stradA, \
                                              ldsBase + (uint32_t)((PB) * 4 * kHalfBytes), wave);                  \
                if (fixB) x_rag_fix<LB, 2, 1>(p.gN, p.gK, (uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)), n0, kT0, kValid, stradB, \
                                              ldsBase + (uint32_t)(((PB) * 4 + 2) * kHalfBytes), wave);            \
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
                __builtin_amdgcn_s_barrier();                                                                      \
            }                                                                                                      \
        }                                                                                                          \
    }

    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t waveLds = VOdometer::sgpr(ldsBase + (uint32_t)wave * 1024u);
    // fragment-read address registers: K-contiguous operand [buffer][k-step] (immediate 2048 x fragment), free-contiguous operand
    // [buffer][fragment] (immediate 8192 x k-step)
    constexpr int nRdA = (LA == LAY_K) ? 2 : 8, nRdB = (LB == LAY_K) ? 2 : 8;
    uint32_t rdA[2][nRdA], rdB[2][nRdB];
#pragma unroll
    for (int P = 0; P < 2; ++P) {
#pragma unroll
        for (int x = 0; x < nRdA; ++x) {
            rdA[P][x] = ldsBase + (uint32_t)((P * 4 + wr) * kHalfBytes) + (LA == LAY_K ? x_offK(lane, x) : x_offF(lane, x));
            asm volatile("" : "+v"(rdA[P][x]));
        }
#pragma unroll
        for (int x = 0; x < nRdB; ++x) {
            rdB[P][x] = ldsBase + (uint32_t)((P * 4 + 2 + wc) * kHalfBytes) + (LB == LAY_K ? x_offK(lane, x) : x_offF(lane, x));
            asm volatile("" : "+v"(rdB[P][x]));
        }
    }

#define CTAMD_X_DMA(P, N, PAD)                                                                                      \
    {                                                                                                              \
        constexpr int q_ = (N) >> 2, i_ = (N) & 3;                                                                 \
        constexpr uint32_t imm_ = (uint32_t)(((P) * 4 + q_) * kHalfBytes + i_ * 4096);                             \
        if constexpr (q_ < 2) v_dma16<imm_, PAD>(v_rsrc<RAG>(odo.addrA), oa.src[q_][i_], waveLds);                 \
        else v_dma16<imm_, PAD>(v_rsrc<RAG>(odo.addrB), ob.src[q_ - 2][i_], waveLds);                              \
    }
#define CTAMD_X_DMA8(P, N0, PAD)                                                                                    \
    CTAMD_X_DMA(P, (N0) + 0, PAD) CTAMD_X_DMA(P, (N0) + 1, PAD) CTAMD_X_DMA(P, (N0) + 2, PAD) CTAMD_X_DMA(P, (N0) + 3, PAD) \
    CTAMD_X_DMA(P, (N0) + 4, PAD) CTAMD_X_DMA(P, (N0) + 5, PAD) CTAMD_X_DMA(P, (N0) + 6, PAD) CTAMD_X_DMA(P, (N0) + 7, PAD)

    // ---- prologue: K-tiles 0 and 1; the odometer stays on tile 1 (k-step 0 of tile t moves it to tile t + 2) -------------
    CTAMD_X_RAGMASK(0)
    CTAMD_X_DMA8(0, 0, true) CTAMD_X_DMA8(0, 8, true)
    odo.advance_a(); odo.advance_b(); odo.advance_event(p.gK);
    CTAMD_X_RAGMASK(1)
    CTAMD_X_DMA8(1, 0, true) CTAMD_X_DMA8(1, 8, true)
    CTAMD_H_VMCNT(16);                            // this wave's pieces of tile 0
    __builtin_amdgcn_s_barrier();
    CTAMD_X_RAGFIX(0, 0)

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 a[2][8], b[2][8];                       // two register sets: k-step s uses set s

    // fragment Q = 0..15 of k-step S from buffer P into register set S: Q < 8 -> B columns 16 Q, else A rows 16 (Q - 8)
#define CTAMD_X_READ(P, S, Q)                                                                                       \
    {                                                                                                              \
        if constexpr ((Q) < 8) {                                                                                   \
            if constexpr (LB == LAY_K) b[S][Q] = v_read<LAY_K, 2048 * (Q)>(rdB[P][(S) % nRdB]);                    \
            else b[S][Q] = v_read<LAY_F, 8192 * (S)>(rdB[P][(Q) % nRdB]);                                          \
        } else {                                                                                                   \
            if constexpr (LA == LAY_K) a[S][(Q) - 8] = v_read<LAY_K, 2048 * ((Q) - 8)>(rdA[P][(S) % nRdA]);        \
            else a[S][(Q) - 8] = v_read<LAY_F, 8192 * (S)>(rdA[P][((Q) - 8) % nRdA]);                              \
        }                                                                                                          \
    }
// MFMA order inside a k-step: the 8 x 8 fragment grid row by row (A fragment fixed for eight MFMAs), even rows walking their columns
// backwards — a row change keeps the B operand of the previous MFMA, so exactly one source operand toggles per MFMA.  Under the
// power limit that is worth +0.3 ... +0.7 percent at 8192^3 on U(-1,1) data (three alternating pairs of runs per layout on one box,
// nothing on zeros: profiles/r05o_mfma_order_ab.jsonl; the mirrored walk — B fragment fixed, A walking — measures the same:
// profiles/r05q_mfma_order_b_stationary_ab.jsonl); -DCTAMD_MFMA_ROWMAJOR restores the plain row-major order for comparison.
#if defined(CTAMD_MFMA_ROWMAJOR)
#define CTAMD_X_MFMA(S, M) x_mfma<BF>(acc[(M) >> 3][(M) & 7], a[S][(M) >> 3], b[S][(M) & 7]);
#else
#define CTAMD_X_MFMA(S, M) x_mfma<BF>(acc[(M) >> 3][(((M) >> 3) & 1) ? ((M) & 7) : 7 - ((M) & 7)], a[S][(M) >> 3], b[S][(((M) >> 3) & 1) ? ((M) & 7) : 7 - ((M) & 7)]);
#endif
    // k-step 0, group Q: one read of k-step 1 (same buffer) and four MFMAs; three of the groups carry the odometer
#define CTAMD_X_G0(P, Q)                                                                                            \
    if constexpr (!kEarly) CTAMD_X_READ(P, 1, Q)                                                                   \
    if constexpr (kEarly && (Q) < 8) { CTAMD_X_READ(P, 1, 2 * ((Q) & 7)) CTAMD_X_READ(P, 1, 2 * ((Q) & 7) + 1) }   \
    CTAMD_X_MFMA(0, 4 * (Q)) CTAMD_X_MFMA(0, 4 * (Q) + 1)                                                          \
    if constexpr ((Q) == 2) odo.advance_a();                                                                       \
    if constexpr ((Q) == 5) odo.advance_b();                                                                       \
    if constexpr ((Q) == 8) odo.advance_event(p.gK);                                                               \
    CTAMD_X_MFMA(0, 4 * (Q) + 2) CTAMD_X_MFMA(0, 4 * (Q) + 3)                                                      \
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1 (behind the barrier), group Q: one read of the next tile's k-step 0 (other buffer), one piece of tile t + 2 into
    // this buffer, four MFMAs
#define CTAMD_X_G1(P, Q)                                                                                            \
    if constexpr (!kEarly) CTAMD_X_READ((P) ^ 1, 0, Q)                                                             \
    if constexpr (kEarly && (Q) < 8) { CTAMD_X_READ((P) ^ 1, 0, 2 * ((Q) & 7)) CTAMD_X_READ((P) ^ 1, 0, 2 * ((Q) & 7) + 1) } \
    CTAMD_X_MFMA(1, 4 * (Q)) CTAMD_X_MFMA(1, 4 * (Q) + 1)                                                          \
    CTAMD_X_DMA(P, Q, false)                                                                                       \
    CTAMD_X_MFMA(1, 4 * (Q) + 2) CTAMD_X_MFMA(1, 4 * (Q) + 3)                                                      \
    __builtin_amdgcn_sched_barrier(0);
#define CTAMD_X_TILE(P)                                                                                             \
    CTAMD_X_G0(P, 0) CTAMD_X_G0(P, 1) CTAMD_X_G0(P, 2) CTAMD_X_G0(P, 3) CTAMD_X_G0(P, 4) CTAMD_X_G0(P, 5)          \
    CTAMD_X_G0(P, 6) CTAMD_X_G0(P, 7) CTAMD_X_G0(P, 8) CTAMD_X_G0(P, 9) CTAMD_X_G0(P, 10) CTAMD_X_G0(P, 11)        \
    CTAMD_X_G0(P, 12) CTAMD_X_G0(P, 13) CTAMD_X_G0(P, 14) CTAMD_X_G0(P, 15)                                        \
    CTAMD_X_RAGMASK(t + (P) + 2)     /* behind the odometer (k-step 0), in front of the pieces of tile t + 2 (k-step 1) */ \
    CTAMD_H_LGKM0();                                                                                               \
    if constexpr (XST != 8 && XST != 10) CTAMD_H_VMCNT(0);                                                         \
    if constexpr (XST != 9 && XST != 10) __builtin_amdgcn_s_barrier();                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_X_RAGFIX(t + (P) + 1, (P) ^ 1)                                                                           \
    CTAMD_X_G1(P, 0) CTAMD_X_G1(P, 1) CTAMD_X_G1(P, 2) CTAMD_X_G1(P, 3) CTAMD_X_G1(P, 4) CTAMD_X_G1(P, 5)          \
    CTAMD_X_G1(P, 6) CTAMD_X_G1(P, 7) CTAMD_X_G1(P, 8) CTAMD_X_G1(P, 9) CTAMD_X_G1(P, 10) CTAMD_X_G1(P, 11)        \
    CTAMD_X_G1(P, 12) CTAMD_X_G1(P, 13) CTAMD_X_G1(P, 14) CTAMD_X_G1(P, 15)

    // first fragments of tile 0
    CTAMD_X_READ(0, 0, 0) CTAMD_X_READ(0, 0, 1) CTAMD_X_READ(0, 0, 2) CTAMD_X_READ(0, 0, 3)
    CTAMD_X_READ(0, 0, 4) CTAMD_X_READ(0, 0, 5) CTAMD_X_READ(0, 0, 6) CTAMD_X_READ(0, 0, 7)
    CTAMD_X_READ(0, 0, 8) CTAMD_X_READ(0, 0, 9) CTAMD_X_READ(0, 0, 10) CTAMD_X_READ(0, 0, 11)
    CTAMD_X_READ(0, 0, 12) CTAMD_X_READ(0, 0, 13) CTAMD_X_READ(0, 0, 14) CTAMD_X_READ(0, 0, 15)
    if constexpr (TIMED) wgStamp[1] = __builtin_readcyclecounter();
    int t = 0;
    for (; t + 1 < nTiles; t += 2) { CTAMD_X_TILE(0) CTAMD_X_TILE(1) }
    if (t < nTiles) { CTAMD_X_TILE(0) }
    CTAMD_H_VMCNT(0);                             // the re-staged tail: no LDS-DMA may outlive the workgroup
    x_acc_ready(acc);
    if constexpr (TIMED) wgStamp[2] = __builtin_readcyclecounter();
    // the lane index again, from the hardware: nothing lane-derived stays live across the main loop for the epilogue's sake (one
    // spilled register = a scratch allocation at every dispatch)
    const int laneE = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));

    const uint32_t mW = m0 + 128 * wr, nW = n0 + 128 * wc;    // this wave's quadrant
    GettParams pe;                                // the epilogue's arguments in one burst of scalar loads (gett_h16w4q_kernel)
    h_reload_params(pe);
    // accumulator fragment (i, j): element r of laneE = row 16 i + 4 (laneE >> 4) + r, column 16 j + (laneE & 15)
    if (pe.partial != nullptr) {                   // split-K: fp32 partial tile, row-major [slice][l][m][n]
        const uint32_t Mt = pe.gM.total, Nt = pe.gN.total;
        float* P = pe.partial + ((size_t)slice * pe.gL.total + l) * (size_t)Mt * Nt;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t m = mW + 16 * i + 4 * (laneE >> 4) + r;
                if (m < Mt) {
                    float* row = P + (size_t)m * Nt;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t n = nW + 16 * j + (laneE & 15);
                        if (n < Nt) row[n] = acc[i][j][r];
                    }
                }
            }
        return;
    }
    __syncthreads();                              // every wave has finished reading the operand ring
    HEpilogue ep;
    ep.init(pe, l, lds, wave);
    unsigned long long tInit = 0, tWrite = 0;     // XST = 7 (measurement): cycles in HEpilogue::init and in the LDS-write phases
    if constexpr (TIMED && XST == 7) tInit = __builtin_readcyclecounter() - wgStamp[2];
    if (XST != 3 && ep.vecD && ep.beta == 0.f) {
        // beta == 0 and 16-byte lanes in D: the accumulators are rounded ONCE on their way into LDS (alpha * acc -> 16 bit), a pass of
        // 32 rows x 128 columns is an image of 272-byte rows (16 bytes of padding: the 2-byte writes of a 16-lane group and the
        // 16-byte reads of a row both spread over the banks), and the way out is eight 16-byte reads + nontemporal stores per lane —
        // whole 256-byte row segments, no conversion behind the LDS.  Half the instructions of the fp32 image below.
        uint16_t* stage = reinterpret_cast<uint16_t*>(ep.scratch);
        constexpr int kPitch = 136;               // 16-bit elements per image row
        const bool interior = ep.flat && mW + 128u <= ep.Mtot && nW + 128u <= ep.Ntot;
        const int64_t stepI = 4 * pe.gM.stride[1][0];                  // four rows on (vecD: the N mode's stride is 1)
        uint16_t* dstI = ep.D + (int64_t)(mW + (uint32_t)(laneE >> 4)) * pe.gM.stride[1][0] + (int64_t)(nW + 8u * (uint32_t)(laneE & 15));
        if (interior) {
            // The whole quadrant inside D and one M / one N mode (wave-uniform).  Software pipeline over the four passes of 32 rows: the
            // eight 16-byte chunks of pass i - 1 (read out of the image into registers) are stored one per eight accumulator elements
            // of pass i on their way into the image — the conversions + 2-byte LDS writes of a pass (1.55k cycles of VALU issue) and its
            // stores (1.65k cycles when all 256 CUs store at once) overlap instead of adding up; no bounds test, addresses one
            // addition apart.  (profiles/r04z_w4x_epilogue_breakdown.jsonl: 16.6k cycles before.)
            s16x8 v[2][8];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {         // group g: fragments (a2 = g >> 2, j = 2 (g & 3) + {0, 1}) of pass i, chunk g of pass i - 1
                    if (i < 4) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int a2 = g >> 2, j = 2 * (g & 3) + jj;
                            const f32x4& c = acc[2 * (i < 4 ? i : 0) + a2][j];
                            uint16_t* st = stage + (16 * a2 + 4 * (laneE >> 4)) * kPitch + 16 * j + (laneE & 15);
                            st[0] = h_round16<BF>(ep.alpha * c[0]); st[kPitch] = h_round16<BF>(ep.alpha * c[1]);
                            st[2 * kPitch] = h_round16<BF>(ep.alpha * c[2]); st[3 * kPitch] = h_round16<BF>(ep.alpha * c[3]);
                        }
                    }
                    if (i > 0) {
                        if constexpr (XST == 1) *reinterpret_cast<s16x8*>(dstI) = v[(i - 1) & 1][g];
                        else __builtin_nontemporal_store(v[(i - 1) & 1][g], reinterpret_cast<s16x8*>(dstI));
                        dstI += stepI;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (i < 4) {
#pragma unroll
                    for (int it = 0; it < 8; ++it)
                        v[i & 1][it] = *reinterpret_cast<const s16x8*>(stage + (4 * it + (laneE >> 4)) * kPitch + 8 * (laneE & 15));
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned long long tw0 = 0;
            if constexpr (TIMED && XST == 7) tw0 = __builtin_readcyclecounter();
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4& c = acc[2 * i + a2][j];
                    uint16_t* st = stage + (16 * a2 + 4 * (laneE >> 4)) * kPitch + 16 * j + (laneE & 15);
                    st[0] = h_round16<BF>(ep.alpha * c[0]); st[kPitch] = h_round16<BF>(ep.alpha * c[1]);
                    st[2 * kPitch] = h_round16<BF>(ep.alpha * c[2]); st[3 * kPitch] = h_round16<BF>(ep.alpha * c[3]);
                }
            if constexpr (TIMED && XST == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tWrite += __builtin_readcyclecounter() - tw0; }
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int q = it * 64 + laneE, row = q >> 4, cc = q & 15;
                const s16x8 v = *reinterpret_cast<const s16x8*>(stage + row * kPitch + 8 * cc);
                const uint32_t m = mW + 32 * i + row, n = nW + 8 * cc;
                if (m < ep.Mtot && n < ep.Ntot) {
                    int64_t offD, offC;
                    ep.offsets(pe, m, n, offD, offC);
                    if constexpr (XST == 1) *reinterpret_cast<s16x8*>(ep.D + offD) = v;
                    else ep.store16(ep.D + offD, v, n);
                }
            }
        }
        }
    } else if (XST == 0 && ep.vecD && ep.vecC && ep.beta != 0.f) {
        // beta != 0 with 16-byte lanes in C and D: the fp32 image of the general path below (alpha acc + beta C is rounded ONCE), as a
        // software pipeline over the four passes of 32 rows.  vmcnt counts loads and stores in issue order, so a wave that asks for the C
        // chunks of a pass BEHIND the stores of the previous pass waits for those stores to be acknowledged before it may use the
        // chunks — four write latencies per tile on top of four read latencies (8192^3: 1.40-1.43 PFLOP/s against 1.60 with beta = 0).
        // Here the chunks of pass i + 1 are requested IN FRONT of the stores of pass i (two named register sets of eight chunks).
#define CTAMD_X_PARK_F32(I)                                                                                                         \
        _Pragma("unroll") for (int F = 0; F < 4; ++F)                                                                              \
            _Pragma("unroll") for (int h = 0; h < 4; ++h) {       /* 16 x 16 quarter (h >> 1, h & 1) of 32 x 32 fragment F */      \
                float* st = ep.scratch + F * 1024 + (16 * (h >> 1) + 4 * (laneE >> 4)) * 32 + 16 * (h & 1) + (laneE & 15);         \
                const f32x4& c = acc[2 * (I) + (h >> 1)][2 * F + (h & 1)];                                                         \
                st[0] = ep.alpha * c[0]; st[32] = ep.alpha * c[1]; st[64] = ep.alpha * c[2]; st[96] = ep.alpha * c[3];             \
            }
#define CTAMD_X_LOADC(I, R)                                                                                                         \
        ep.template load_c_chunk<0>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##0); ep.template load_c_chunk<1>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##1); \
        ep.template load_c_chunk<2>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##2); ep.template load_c_chunk<3>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##3); \
        ep.template load_c_chunk<4>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##4); ep.template load_c_chunk<5>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##5); \
        ep.template load_c_chunk<6>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##6); ep.template load_c_chunk<7>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##7);
#define CTAMD_X_COMBINE(I, R)                                                                                                       \
        ep.template store_chunk_with_c<BF, 0>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##0); ep.template store_chunk_with_c<BF, 1>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##1); \
        ep.template store_chunk_with_c<BF, 2>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##2); ep.template store_chunk_with_c<BF, 3>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##3); \
        ep.template store_chunk_with_c<BF, 4>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##4); ep.template store_chunk_with_c<BF, 5>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##5); \
        ep.template store_chunk_with_c<BF, 6>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##6); ep.template store_chunk_with_c<BF, 7>(pe, mW + 32u * (I), nW, 64u, 32u, laneE, R##7);
        s16x8 ca0 = {}, ca1 = {}, ca2 = {}, ca3 = {}, ca4 = {}, ca5 = {}, ca6 = {}, ca7 = {};
        s16x8 cb0 = {}, cb1 = {}, cb2 = {}, cb3 = {}, cb4 = {}, cb5 = {}, cb6 = {}, cb7 = {};
        CTAMD_X_LOADC(0, ca)
        CTAMD_X_PARK_F32(0) CTAMD_X_LOADC(1, cb) CTAMD_X_COMBINE(0, ca)
        CTAMD_X_PARK_F32(1) CTAMD_X_LOADC(2, ca) CTAMD_X_COMBINE(1, cb)
        CTAMD_X_PARK_F32(2) CTAMD_X_LOADC(3, cb) CTAMD_X_COMBINE(2, ca)
        CTAMD_X_PARK_F32(3) CTAMD_X_COMBINE(3, cb)
#undef CTAMD_X_PARK_F32
#undef CTAMD_X_LOADC
#undef CTAMD_X_COMBINE
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {                 // four passes of 32 rows: the epilogue's image is four 32 x 32 fp32 fragments
#pragma unroll
        for (int F = 0; F < 4; ++F)
#pragma unroll
            for (int h = 0; h < 4; ++h) {         // 16 x 16 quarter (h >> 1, h & 1) of 32 x 32 fragment F
                float* st = ep.scratch + F * 1024 + (16 * (h >> 1) + 4 * (laneE >> 4)) * 32 + 16 * (h & 1) + (laneE & 15);
                const f32x4& c = acc[2 * i + (h >> 1)][2 * F + (h & 1)];
                st[0] = ep.alpha * c[0]; st[32] = ep.alpha * c[1]; st[64] = ep.alpha * c[2]; st[96] = ep.alpha * c[3];
            }
        const uint32_t mB = mW + 32 * i;
        ep.template flush<BF, (XST == 1 ? 1 : 0)>(pe, mB, 0u, 0u, nW, 64u, 32u, laneE);
    }
    if constexpr (TIMED) {
        if (p.timing != nullptr && wave == 0 && laneE == 0) {
            wgStamp[3] = __builtin_readcyclecounter();            // the stores are issued, not waited for
            wgStamp[5] = wall_clock64();
#pragma unroll
            for (int i = 0; i < 6; ++i) p.timing[64 + 8 * (size_t)blockIdx.x + i] = wgStamp[i];
            p.timing[64 + 8 * (size_t)blockIdx.x + 6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;   // HW_REG_XCC_ID
            if constexpr (XST == 7) { p.timing[64 + 8 * (size_t)blockIdx.x + 1] = tInit; p.timing[64 + 8 * (size_t)blockIdx.x + 7] = tWrite; }
        }
    }
}

template <bool BF, int LA, int LB>
static hipError_t launch_h16w4x(const GettParams& p, hipStream_t stream) {
    const bool needRag = p.gK.total % (uint32_t)kHBK != 0u || (p.ragged & 1u) != 0u;   // (the TIMED / XST instantiations have no masked tile)
#if defined(CTAMD_RESEARCH_KERNELS)
    if constexpr (BF && LA == LAY_K && LB == LAY_F) if (!needRag) {   // the one instantiation that carries the in-kernel timestamps / store modes
        static const bool timed = [] { const char* e = getenv("CUTENSOR_AMD_H16_TIMED"); return e && e[0] == '1'; }();
        static const int xst = [] { const char* e = getenv("CUTENSOR_AMD_H16_XST"); return e ? atoi(e) : 0; }();
        if (timed && xst == 1) { hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, true, 1>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (timed && xst == 3) { hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, true, 3>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (timed && xst == 4) { hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, true, 4>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (timed && xst == 5) { hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, true, 5>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (timed && xst == 7) { hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, true, 7>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (timed && xst == 8) { hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, true, 8>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (timed && xst == 9) { hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, true, 9>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (timed && xst == 10) { hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, true, 10>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (timed) { hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, true>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
    }
#endif
    if (needRag) {   // ragged K (one contracted mode) or a unit that can straddle the tensor's end (pick_h16_choice): the masked last K-tile
        hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB, false, 0, true>), dim3(p.nBlocks), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((gett_h16w4x_kernel<BF, LA, LB>), dim3(p.nBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}


// =====================================================================================================
// gett_h16w4m_kernel (CUTENSOR_AMD_H16_WAVES=4m): the 128 x 128 x 64 sibling of gett_h16w4x_kernel for MID-SIZE problems.
// One 256 x 256 tile per CU leaves a 2048^3 problem on 64 of the 256 CUs (or pays a split-K fold), and a workgroup of the large
// kernels spends ~11 us outside its main loop whatever K is.  Here: the same instruction (v_mfma_f32_16x16x32 from inline asm),
// the same LDS images / source-side swizzles / LDS-DMA staging / K odometer, but a 64 x 64 quadrant per wave = 4 x 4 accumulator
// fragments (64 AGPRs), ONE 128-row half-tile per operand and K-tile (32 KiB), two K-tiles in LDS (64 KiB) — so TWO workgroups
// share a CU (launch bounds 256 x 2): one workgroup's prologue, tile barrier and epilogue sit under the other's MFMAs, which is
// what the persistent-workgroup form of the large kernel would have to build by hand.  Per K-tile and wave: 32 MFMAs, 16 fragment
// reads (one per two MFMAs, into the other of two register sets), 8 LDS-DMA pieces (one per two MFMAs of k-step 1, tile t + 2).
// Twice the LDS-DMA bytes per flop of the 256 x 256 tile: the kernel for problems whose 256 x 256 tiles do not fill the chip, not
// for the 8192^3 class.  Epilogue: the 16-bit image of gett_h16w4x_kernel in passes of 32 rows x 64 columns (beta = 0, 16-byte
// lanes in D), else one pass of four 32 x 32 fp32 fragments through HEpilogue::flush.
// =====================================================================================================
//
// R = ring depth in K-tiles (R = 4 also spreads the LDS-DMA pieces of a tile evenly over a K-tile time: one piece per four MFMAs
// in both k-steps instead of one per two MFMAs in k-step 1 only — issued in bursts the pieces ask the texture path for twice its
// 64 B/clk and the MFMAs behind them wait: ~1000 cycles per K-tile measured with the burst, profiles/r04d_*).  R = 2 (64 KiB, two workgroups per CU): tile t + 2 is staged during k-step 1 of tile t and must have
// landed one and a half K-tiles (~770 MFMA cycles) later — enough when a second workgroup fills the gaps, not for a workgroup that
// has its CU to itself (2048^3: 1430 cycles per K-tile against 512 of MFMA issue, profiles/r04c_h16_shape_sweep.txt).  R = 4
// (128 KiB, one workgroup per CU): tile t + 4 is staged during tile t, the tile barrier waits for tile t + 1 only (counted
// vmcnt: the 16 pieces of tiles t + 2, t + 3 stay in flight) — the form for problems with at most one 128 x 128 tile per CU.
constexpr int kMTile = 128;
// RAG: ragged K, as in gett_h16w4x_kernel (x_rag_mask).  R = 4 stages a tile's A and B pieces at different times: one switch each.
template <bool BF, int LA, int LB, int R = 2, bool RAG = false>
__global__ void __launch_bounds__(256, (R == 2 ? 2 : 1)) gett_h16w4m_kernel(const GettParams p) {
    static_assert(R == 2 || R == 4, "ring of two or four K-tiles");
    __shared__ __attribute__((aligned(16))) char lds[R * 2 * kHalfBytes];       // buffer P: [A half-tile][B half-tile]
    prefetch_kernarg<(int)sizeof(GettParams)>();
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    uint32_t id = xcd_remap(blockIdx.x, p.nBlocks);
    const uint32_t tilesMN = p.tilesM * p.tilesN;
    const uint32_t tilesAll = tilesMN * p.gL.total;
    const uint32_t slice = id / tilesAll;
    id -= slice * tilesAll;
    const uint32_t l = id / tilesMN;
    id -= l * tilesMN;
    const uint32_t perGroup = 8u * p.tilesN;
    const uint32_t grp = id / perGroup, inGrp = id - grp * perGroup;
    const uint32_t first = grp * 8u;
    const uint32_t gsz = (p.tilesM - first < 8u) ? (p.tilesM - first) : 8u;
    const uint32_t mt = first + inGrp % gsz, nt = inGrp / gsz;
    const uint32_t m0 = p.mOrg + mt * kMTile, n0 = p.nOrg + nt * kMTile;
    const bool sweep = RAG && (p.ragged & 2u) != 0u;          // sweep-ragged K: gett_h16w4x_kernel
    const uint32_t kTilesAll = sweep ? (p.ragged >> 2) : (p.gK.total + (RAG ? (uint32_t)kHBK - 1u : 0u)) / kHBK, tilesPerSlice = p.kPerSlice / kHBK;
    const uint32_t tile0 = slice * tilesPerSlice;
    const int nTiles = (int)((tile0 + tilesPerSlice <= kTilesAll) ? tilesPerSlice : (kTilesAll - tile0));

    HOperand<LA, 4, false, 1, 1> oa;
    HOperand<LB, 4, false, 1, 1> ob;
    oa.init(p.gM, p.gK.stride[0][0], m0, wave, lane);
    ob.init(p.gN, p.gK.stride[1][0], n0, wave, lane);
    const uint64_t bA = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)) + oa.base);
    const uint64_t bB = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)) + ob.base);
    VOdometer odo;
    if (sweep) odo.init_tiles(p.gK, tile0, (uint32_t)nTiles, bA, bB);
    else odo.template init<RAG>(p.gK, tile0 * kHBK, (uint32_t)nTiles, bA, bB);
    // ragged K / operands without 16-byte lanes: index (among this workgroup's K-tiles) of the tile that is staged masked — the last K-tile
    // of the last slice — and how many k of the contracted range it holds (x_rag_mask, x_rag_fix: gett_h16x_common.h)
    const int maskAt = (RAG && !sweep && tile0 + (uint32_t)nTiles == kTilesAll) ? nTiles - 1 : 0x7fffffff;
    const uint32_t kValid = VOdometer::sgpr(sweep ? p.gK.div[0].d - (odo.n0 - 1u) * (uint32_t)kHBK
                                                  : ((p.gK.total % kHBK) != 0u ? p.gK.total % kHBK : (uint32_t)kHBK));
    uint32_t stradA = 0u, stradB = 0u;            // units of the masked tile that x_rag_fix loads element by element
    bool maskOnA = false, maskOnB = false;        // sweep-ragged K: one switch per operand (the deep ring stages A and B of a tile at different times)
    // called with the operand's descriptor base ON tile IDX (sweep: tiles past the slice's last one are re-staged copies — the mask stays)
#define CTAMD_M_RAGMASK_A(IDX) if constexpr (RAG) {                                                                    \
        if ((IDX) == maskAt) stradA = x_rag_mask<LA, 1>(oa.src, wave, kValid, x_rag_limit(p.endA, odo.addrA));        \
        if (sweep && (IDX) < nTiles && odo.on_sweep_end() != maskOnA) { x_rag_toggle<LA, 1>(oa.src, wave, kValid); maskOnA = !maskOnA; } }
#define CTAMD_M_RAGMASK_B(IDX) if constexpr (RAG) {                                                                    \
        if ((IDX) == maskAt) stradB = x_rag_mask<LB, 1>(ob.src, wave, kValid, x_rag_limit(p.endB, odo.addrB));        \
        if (sweep && (IDX) < nTiles && odo.on_sweep_end() != maskOnB) { x_rag_toggle<LB, 1>(ob.src, wave, kValid); maskOnB = !maskOnB; } }
    // tile IDX (the masked one) has landed in buffer PB, behind a workgroup barrier (gett_h16w4x_kernel, CTAMD_X_RAGFIX)
#define CTAMD_M_RAGFIX(IDX, PB)                                                                                     \
    if constexpr (RAG) {                                                                                           \
        if ((IDX) == maskAt) {                                                                                     \
            const bool fixA = (LA == LAY_K) ? (kValid & 7u) != 0u : ((p.gM.total & 7u) != 0u && m0 + (uint32_t)kMTile >= p.gM.total);   \
            const bool fixB = (LB == LAY_K) ? (kValid & 7u) != 0u : ((p.gN.total & 7u) != 0u && n0 + (uint32_t)kMTile >= p.gN.total);   \
            if (fixA || fixB) {                                                                                    \
                const uint32_t kT0 = (kTilesAll - 1u) * (uint32_t)kHBK;                                            \
                if (fixA) x_rag_fix<LA, 1, 0>(p.gM, p.gK, (uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)), m0, kT0, kValid, stradA, \
                                              ldsBase + (uint32_t)((PB) * 2 * kHalfBytes), wave);                  \
                if (fixB) x_rag_fix<LB, 1, 1>(p.gN, p.gK, (uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)), n0, kT0, kValid, stradB, \
                                              ldsBase + (uint32_t)(((PB) * 2 + 1) * kHalfBytes), wave);            \
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
                __builtin_amdgcn_s_barrier();                                                                      \
            }                                                                                                      \
        }                                                                                                          \
    }

    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t waveLds = VOdometer::sgpr(ldsBase + (uint32_t)wave * 1024u);
    // fragment-read address registers.  This wave's fragments are f = 4 wr + i (A) / 4 wc + j (B) of the 128-row half-tile.
    // K-contiguous operand: [buffer][k-step], the fragment in the immediate (2048 x i on top of 8192 x wr in the register);
    // free-contiguous operand: [buffer][fragment] (the fragment index sits inside the swizzle), the k-step in the immediate (8192 x s)
    constexpr int nRdA = (LA == LAY_K) ? 2 : 4, nRdB = (LB == LAY_K) ? 2 : 4;
    uint32_t rdA[R][nRdA], rdB[R][nRdB];
#pragma unroll
    for (int P = 0; P < R; ++P) {
#pragma unroll
        for (int x = 0; x < nRdA; ++x) {
            rdA[P][x] = ldsBase + (uint32_t)((2 * P) * kHalfBytes) + (LA == LAY_K ? (uint32_t)(8192 * wr) + x_offK(lane, x) : x_offF(lane, 4 * wr + x));
            asm volatile("" : "+v"(rdA[P][x]));
        }
#pragma unroll
        for (int x = 0; x < nRdB; ++x) {
            rdB[P][x] = ldsBase + (uint32_t)((2 * P + 1) * kHalfBytes) + (LB == LAY_K ? (uint32_t)(8192 * wc) + x_offK(lane, x) : x_offF(lane, 4 * wc + x));
            asm volatile("" : "+v"(rdB[P][x]));
        }
    }

    // piece N = 0..7 of the K-tile the odometer describes, into buffer P: operand q = N >> 2 (A, B), piece i = N & 3 of this wave
#define CTAMD_M_DMA(P, N, PAD)                                                                                      \
    {                                                                                                              \
        constexpr int q_ = (N) >> 2, i_ = (N) & 3;                                                                 \
        constexpr uint32_t imm_ = (uint32_t)(((P) * 2 + q_) * kHalfBytes + i_ * 4096);                             \
        if constexpr (q_ == 0) v_dma16<imm_, PAD>(v_rsrc<RAG>(odo.addrA), oa.src[0][i_], waveLds);                 \
        else v_dma16<imm_, PAD>(v_rsrc<RAG>(odo.addrB), ob.src[0][i_], waveLds);                                   \
    }
#define CTAMD_M_DMA8(P, PAD)                                                                                        \
    CTAMD_M_DMA(P, 0, PAD) CTAMD_M_DMA(P, 1, PAD) CTAMD_M_DMA(P, 2, PAD) CTAMD_M_DMA(P, 3, PAD)                    \
    CTAMD_M_DMA(P, 4, PAD) CTAMD_M_DMA(P, 5, PAD) CTAMD_M_DMA(P, 6, PAD) CTAMD_M_DMA(P, 7, PAD)

    // ---- prologue: K-tiles 0 .. R - 1; the odometer stays on tile R - 1 (k-step 0 of tile t moves it to tile t + R) ---------
    CTAMD_M_RAGMASK_A(0) CTAMD_M_RAGMASK_B(0)
    CTAMD_M_DMA8(0, true)
    odo.advance_a(); odo.advance_b(); odo.advance_event(p.gK);
    CTAMD_M_RAGMASK_A(1) CTAMD_M_RAGMASK_B(1)
    CTAMD_M_DMA8(1, true)
    if constexpr (R == 4) {
        // the deep ring spreads a tile's eight pieces over one K-tile time (four behind the barrier of tile t, four in front of the
        // barrier of tile t + 1): tiles 0 .. 2 and the first half of tile 3 here, its second half rides in k-step 0 of tile 0
        odo.advance_a(); odo.advance_b(); odo.advance_event(p.gK);
        CTAMD_M_RAGMASK_A(2) CTAMD_M_RAGMASK_B(2)
        CTAMD_M_DMA8(2, true)
        odo.advance_a(); odo.advance_b(); odo.advance_event(p.gK);
        CTAMD_M_RAGMASK_A(3)
        CTAMD_M_DMA(3, 0, true) CTAMD_M_DMA(3, 1, true) CTAMD_M_DMA(3, 2, true) CTAMD_M_DMA(3, 3, true)
    }
    CTAMD_H_VMCNT(R == 2 ? 8 : 20);               // this wave's pieces of tile 0
    __builtin_amdgcn_s_barrier();
    CTAMD_M_RAGFIX(0, 0)

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 a[2][4], b[2][4];                       // two register sets: k-step s uses set s

    // fragment Q = 0..7 of k-step S from buffer P into register set S: Q < 4 -> B columns 16 Q, else A rows 16 (Q - 4)
#define CTAMD_M_READ(P, S, Q)                                                                                       \
    {                                                                                                              \
        if constexpr ((Q) < 4) {                                                                                   \
            if constexpr (LB == LAY_K) b[S][Q] = v_read<LAY_K, 2048 * (Q)>(rdB[P][(S) % nRdB]);                    \
            else b[S][Q] = v_read<LAY_F, 8192 * (S)>(rdB[P][(Q) % nRdB]);                                          \
        } else {                                                                                                   \
            if constexpr (LA == LAY_K) a[S][(Q) - 4] = v_read<LAY_K, 2048 * ((Q) - 4)>(rdA[P][(S) % nRdA]);        \
            else a[S][(Q) - 4] = v_read<LAY_F, 8192 * (S)>(rdA[P][((Q) - 4) % nRdA]);                              \
        }                                                                                                          \
    }
#define CTAMD_M_MFMA(S, M) x_mfma<BF>(acc[(M) >> 2][(M) & 3], a[S][(M) >> 2], b[S][(M) & 3]);
    // k-step 0, group Q: one read of k-step 1 (same buffer) and two MFMAs; three of the groups carry the odometer
    // (R = 4: the groups also carry pieces 4..7 of tile t + 3 into the buffer behind this one, then the odometer moves on)
#define CTAMD_M_G0(P, Q)                                                                                            \
    if constexpr ((Q) < 4) { CTAMD_M_READ(P, 1, 2 * (Q)) CTAMD_M_READ(P, 1, 2 * (Q) + 1) }   /* reads early: see gett_h16w8m_kernel */ \
    CTAMD_M_MFMA(0, 2 * (Q))                                                                                       \
    if constexpr (R == 2) {                                                                                        \
        if constexpr ((Q) == 1) odo.advance_a();                                                                   \
        if constexpr ((Q) == 3) odo.advance_b();                                                                   \
        if constexpr ((Q) == 5) odo.advance_event(p.gK);                                                           \
    } else {                                                                                                       \
        if constexpr ((Q) % 2 == 0) CTAMD_M_DMA(((P) + R - 1) % R, 4 + (Q) / 2, false)                             \
        if constexpr ((Q) == 7) { odo.advance_a(); odo.advance_b(); odo.advance_event(p.gK); }                     \
    }                                                                                                              \
    CTAMD_M_MFMA(0, 2 * (Q) + 1)                                                                                   \
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1 (behind the barrier), group Q: one read of the next tile's k-step 0 (other buffer), one piece of tile t + 2 into this
    // buffer, two MFMAs
#define CTAMD_M_G1(P, Q)                                                                                            \
    if constexpr ((Q) < 4) { CTAMD_M_READ(((P) + 1) % R, 0, 2 * (Q)) CTAMD_M_READ(((P) + 1) % R, 0, 2 * (Q) + 1) } \
    CTAMD_M_MFMA(1, 2 * (Q))                                                                                       \
    if constexpr (R == 2) CTAMD_M_DMA(P, Q, false)                                                                 \
    else if constexpr ((Q) % 2 == 0) CTAMD_M_DMA(P, (Q) / 2, false)         /* pieces 0..3 of tile t + 4 */          \
    CTAMD_M_MFMA(1, 2 * (Q) + 1)                                                                                   \
    __builtin_amdgcn_sched_barrier(0);
    // tile t (buffer P = t mod R): R = 2 stages tile t + 2 in k-step 1; R = 4 the B pieces of tile t + 3 in k-step 0, the A pieces of
    // tile t + 4 in k-step 1 (the ragged-K switches sit right in front of them)
#define CTAMD_M_TILE(P)                                                                                             \
    if constexpr (R == 4) { CTAMD_M_RAGMASK_B(t + (P) + 3) }         /* the odometer is on tile t + 3 until group 7 of k-step 0 */ \
    CTAMD_M_G0(P, 0) CTAMD_M_G0(P, 1) CTAMD_M_G0(P, 2) CTAMD_M_G0(P, 3)                                            \
    CTAMD_M_G0(P, 4) CTAMD_M_G0(P, 5) CTAMD_M_G0(P, 6) CTAMD_M_G0(P, 7)                                            \
    if constexpr (R == 2) { CTAMD_M_RAGMASK_A(t + (P) + 2) CTAMD_M_RAGMASK_B(t + (P) + 2) }     /* behind the odometer of k-step 0 */ \
    CTAMD_H_LGKM0();                                                                                               \
    CTAMD_H_VMCNT(8 * (R - 2));      /* tile t + 1 has landed; the pieces of tiles t + 2 .. t + R - 1 stay in flight */     \
    __builtin_amdgcn_s_barrier();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_M_RAGFIX(t + (P) + 1, ((P) + 1) % R)                                                                     \
    if constexpr (R == 4) { CTAMD_M_RAGMASK_A(t + (P) + 4) }                                                       \
    CTAMD_M_G1(P, 0) CTAMD_M_G1(P, 1) CTAMD_M_G1(P, 2) CTAMD_M_G1(P, 3)                                            \
    CTAMD_M_G1(P, 4) CTAMD_M_G1(P, 5) CTAMD_M_G1(P, 6) CTAMD_M_G1(P, 7)

    // first fragments of tile 0
    CTAMD_M_READ(0, 0, 0) CTAMD_M_READ(0, 0, 1) CTAMD_M_READ(0, 0, 2) CTAMD_M_READ(0, 0, 3)
    CTAMD_M_READ(0, 0, 4) CTAMD_M_READ(0, 0, 5) CTAMD_M_READ(0, 0, 6) CTAMD_M_READ(0, 0, 7)
    int t = 0;
    if constexpr (R == 2) {
        for (; t + 1 < nTiles; t += 2) { CTAMD_M_TILE(0) CTAMD_M_TILE(1) }
        if (t < nTiles) { CTAMD_M_TILE(0) }
    } else {
        for (; t + 3 < nTiles; t += 4) { CTAMD_M_TILE(0) CTAMD_M_TILE(1) CTAMD_M_TILE(2) CTAMD_M_TILE(3) }
        if (t < nTiles) { CTAMD_M_TILE(0) }
        if (t + 1 < nTiles) { CTAMD_M_TILE(1) }
        if (t + 2 < nTiles) { CTAMD_M_TILE(2) }
    }
    CTAMD_H_VMCNT(0);                             // the re-staged tail: no LDS-DMA may outlive the workgroup
    x_acc_ready(acc);
    const int laneE = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));

    const uint32_t mW = m0 + 64 * wr, nW = n0 + 64 * wc;      // this wave's quadrant
    GettParams pe;                                // the epilogue's arguments in one burst of scalar loads (gett_h16w4q_kernel)
    h_reload_params(pe);
    // accumulator fragment (i, j): element r of laneE = row 16 i + 4 (laneE >> 4) + r, column 16 j + (laneE & 15)
    if (pe.partial != nullptr) {                   // split-K: fp32 partial tile, row-major [slice][l][m][n]
        const uint32_t Mt = pe.gM.total, Nt = pe.gN.total;
        float* P = pe.partial + ((size_t)slice * pe.gL.total + l) * (size_t)Mt * Nt;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t m = mW + 16 * i + 4 * (laneE >> 4) + r;
                if (m < Mt) {
                    float* row = P + (size_t)m * Nt;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t n = nW + 16 * j + (laneE & 15);
                        if (n < Nt) row[n] = acc[i][j][r];
                    }
                }
            }
        return;
    }
    __syncthreads();                              // every wave has finished reading the operand ring
    HEpilogue ep;
    ep.init(pe, l, lds, wave);                     // 16 KiB of the (dead) ring per wave
    if (ep.vecD && ep.beta == 0.f) {
        // beta == 0 and 16-byte lanes in D: rounded once on the way into a 16-bit image of 32 rows x 64 columns (144-byte rows: the
        // 2-byte writes of a 16-lane group and the 16-byte reads of a row both spread over the banks), out as 16-byte reads +
        // nontemporal stores of whole 128-byte row segments
        uint16_t* stage = reinterpret_cast<uint16_t*>(ep.scratch);
        constexpr int kPitch = 72;                // 16-bit elements per image row
        if (ep.flat && mW + 64u <= ep.Mtot && nW + 64u <= ep.Ntot) {      // interior quadrant: pipelined, no bounds tests
            const int64_t sM = pe.gM.stride[1][0];
            x_store_interior16<BF, 4, 4, kPitch>(acc, stage, ep.alpha,
                                                 ep.D + (int64_t)(mW + (uint32_t)(laneE >> 3)) * sM + (int64_t)(nW + 8u * (uint32_t)(laneE & 7)), 8 * sM, laneE);
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4& c = acc[2 * i + a2][j];
                    uint16_t* st = stage + (16 * a2 + 4 * (laneE >> 4)) * kPitch + 16 * j + (laneE & 15);
                    st[0] = h_round16<BF>(ep.alpha * c[0]); st[kPitch] = h_round16<BF>(ep.alpha * c[1]);
                    st[2 * kPitch] = h_round16<BF>(ep.alpha * c[2]); st[3 * kPitch] = h_round16<BF>(ep.alpha * c[3]);
                }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = it * 64 + laneE, row = q >> 3, cc = q & 7;
                const s16x8 v = *reinterpret_cast<const s16x8*>(stage + row * kPitch + 8 * cc);
                const uint32_t m = mW + 32 * i + row, n = nW + 8 * cc;
                if (m < ep.Mtot && n < ep.Ntot) {
                    int64_t offD, offC;
                    ep.offsets(pe, m, n, offD, offC);
                    ep.store16(ep.D + offD, v, n);
                }
            }
        }
    } else {
        // one pass: the wave's 64 x 64 quadrant as four 32 x 32 fp32 fragments F = 2 (row block) + (column block)
#pragma unroll
        for (int F = 0; F < 4; ++F)
#pragma unroll
            for (int h = 0; h < 4; ++h) {         // 16 x 16 quarter (h >> 1, h & 1) of fragment F
                float* st = ep.scratch + F * 1024 + (16 * (h >> 1) + 4 * (laneE >> 4)) * 32 + 16 * (h & 1) + (laneE & 15);
                const f32x4& c = acc[2 * (F >> 1) + (h >> 1)][2 * (F & 1) + (h & 1)];
                st[0] = ep.alpha * c[0]; st[32] = ep.alpha * c[1]; st[64] = ep.alpha * c[2]; st[96] = ep.alpha * c[3];
            }
        ep.template flush<BF>(pe, mW, 32u, 0u, nW, 0u, 32u, laneE);
    }
}


// =====================================================================================================
// gett_h16w8m_kernel (CUTENSOR_AMD_H16_WAVES=8m): the 128 x 128 x 64 tile with DEDICATED data-moving waves.
// tools/ubench/ldsdma_rate.hip measures what a CU pulls into LDS through buffer_load ... lds when it does nothing else: ~60 B/clk
// (130 GB/s per CU, 33-35 TB/s chip-wide) from L2, with or without LDS readers beside it — yet every four-wave 16-bit kernel here
// ends up at 28-36 B/clk/CU (profiles/r04f_h16_shape_sweep.txt), and a 128 x 128 x 64 tile needs 64 B/clk to keep its MFMAs fed.
// This kernel tests the ISSUE hypothesis: an LDS-DMA instruction holds its wave's issue port for 60-180 cycles
// (tools/ubench/dma_issue.hip), and a wave that owns a SIMD's matrix pipe cannot queue MFMAs meanwhile.  So, as in the fp32
// streaming kernel (gett_f32_stream.hip): eight waves, waves 0-3 multiply (one per SIMD: the compute stream of gett_h16w4m_kernel
// WITHOUT its LDS-DMA pieces, odometer and vmcnt waits), waves 4-7 only move data (one per SIMD: staging tables, K odometer, eight
// pieces per K-tile each) — a piece's issue time now overlaps the OTHER wave's MFMAs.  R K-tiles of 32 KiB in LDS, one workgroup
// per CU, ONE barrier per K-tile for all eight waves:
//   barrier #0      : tile 0 has landed
//   barrier #(t+1)  : tile t + 1 has landed (every mover waited for its own pieces: vmcnt(8 (R - 2)), the tiles behind it stay in
//                     flight) and every multiplying wave holds the last fragments of tile t in registers -> buffer t % R is
//                     refilled with tile t + R
// Measured (profiles/r04k..r04m): +3 % over the four-wave ring-4 form on 2048^3-class shapes, +10-20 % at 1024^3 (shorter prologue:
// the movers start while the multiplying waves set up), and R = 5 (all 160 KiB) = R = 4 — neither the issue port nor latency x bytes
// in flight is what holds these kernels at ~0.42 us per K-tile (NOTES.md, round 4, "what was ruled out").  An autotuning candidate
// and CUTENSOR_AMD_H16_WAVES=8m; not ranked by the planner.
// Same LDS images, swizzles, fragment addressing and epilogues as gett_h16w4m_kernel.
// =====================================================================================================
template <bool BF, int LA, int LB, int R = 5>
__global__ void __launch_bounds__(512, 1) gett_h16w8m_kernel(const GettParams p) {
    static_assert(R == 4 || R == 5, "ring of four or five K-tiles (128 / 160 KiB)");
    __shared__ __attribute__((aligned(16))) char lds[R * 2 * kHalfBytes];       // buffer P: [A half-tile][B half-tile]
    prefetch_kernarg<(int)sizeof(GettParams)>();
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;
    const bool mover = wave8 >= 4;
    const int wr = wave >> 1, wc = wave & 1;

    uint32_t id = xcd_remap(blockIdx.x, p.nBlocks);
    const uint32_t tilesMN = p.tilesM * p.tilesN;
    const uint32_t tilesAll = tilesMN * p.gL.total;
    const uint32_t slice = id / tilesAll;
    id -= slice * tilesAll;
    const uint32_t l = id / tilesMN;
    id -= l * tilesMN;
    const uint32_t perGroup = 8u * p.tilesN;
    const uint32_t grp = id / perGroup, inGrp = id - grp * perGroup;
    const uint32_t first = grp * 8u;
    const uint32_t gsz = (p.tilesM - first < 8u) ? (p.tilesM - first) : 8u;
    const uint32_t mt = first + inGrp % gsz, nt = inGrp / gsz;
    const uint32_t m0 = mt * kMTile, n0 = nt * kMTile;
    const uint32_t kTilesAll = p.gK.total / kHBK, tilesPerSlice = p.kPerSlice / kHBK;
    const uint32_t tile0 = slice * tilesPerSlice;
    const int nTiles = (int)((tile0 + tilesPerSlice <= kTilesAll) ? tilesPerSlice : (kTilesAll - tile0));
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const bool splitK = p.partial != nullptr;

    if (mover) {
        // =========================== data movers ======================================================
        HOperand<LA, 4, false, 1, 1> oa;
        HOperand<LB, 4, false, 1, 1> ob;
        oa.init(p.gM, p.gK.stride[0][0], m0, wave, lane);
        ob.init(p.gN, p.gK.stride[1][0], n0, wave, lane);
        const uint64_t bA = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)) + oa.base);
        const uint64_t bB = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)) + ob.base);
        VOdometer odo;
        odo.init(p.gK, tile0 * kHBK, (uint32_t)nTiles, bA, bB);
        const uint32_t waveLds = VOdometer::sgpr(ldsBase + (uint32_t)wave * 1024u);
        // the eight pieces of the K-tile the odometer describes into the buffer at LDS byte offset BUFOFF (wave-uniform), then on to
        // the next K-tile (beyond the K range the odometer stays where it is: the last tile is staged again and never read)
#define CTAMD_W_ISSUE(BUFOFF, PAD)                                                                                   \
        {                                                                                                          \
            const uint32_t wl_ = VOdometer::sgpr(waveLds + (BUFOFF));                                              \
            v_dma16<0 * 4096, PAD>(v_rsrc(odo.addrA), oa.src[0][0], wl_);                                          \
            v_dma16<1 * 4096, PAD>(v_rsrc(odo.addrA), oa.src[0][1], wl_);                                          \
            v_dma16<2 * 4096, PAD>(v_rsrc(odo.addrA), oa.src[0][2], wl_);                                          \
            v_dma16<3 * 4096, PAD>(v_rsrc(odo.addrA), oa.src[0][3], wl_);                                          \
            v_dma16<(uint32_t)kHalfBytes + 0 * 4096, PAD>(v_rsrc(odo.addrB), ob.src[0][0], wl_);                   \
            v_dma16<(uint32_t)kHalfBytes + 1 * 4096, PAD>(v_rsrc(odo.addrB), ob.src[0][1], wl_);                   \
            v_dma16<(uint32_t)kHalfBytes + 2 * 4096, PAD>(v_rsrc(odo.addrB), ob.src[0][2], wl_);                   \
            v_dma16<(uint32_t)kHalfBytes + 3 * 4096, PAD>(v_rsrc(odo.addrB), ob.src[0][3], wl_);                   \
            odo.advance_a(); odo.advance_b(); odo.advance_event(p.gK);                                             \
        }
        constexpr uint32_t kBuf = 2u * (uint32_t)kHalfBytes;
        CTAMD_W_ISSUE(0u * kBuf, true)
        CTAMD_W_ISSUE(1u * kBuf, true)
        CTAMD_H_VMCNT(8);                              // progressive start: the multipliers go as soon as tile 0 is there
        __builtin_amdgcn_s_barrier();                  // #0
        CTAMD_W_ISSUE(2u * kBuf, true)
        CTAMD_W_ISSUE(3u * kBuf, true)
        if constexpr (R == 5) CTAMD_W_ISSUE(4u * kBuf, true)
        uint32_t bufOff = 0;
        for (int t = 0; t < nTiles; ++t) {
            CTAMD_H_VMCNT(8 * (R - 2));                // tile t + 1 has landed; tiles t + 2 .. t + R - 1 stay in flight
            __builtin_amdgcn_s_barrier();              // #(t + 1): buffer t % 4 is free
            CTAMD_W_ISSUE(bufOff, false)               // tile t + R
            bufOff = (bufOff + kBuf == (uint32_t)R * kBuf) ? 0u : bufOff + kBuf;
        }
        CTAMD_H_VMCNT(0);                              // no LDS-DMA may outlive the workgroup (or land in the epilogue's scratch)
        if (!splitK) __builtin_amdgcn_s_barrier();     // the epilogue's barrier: every piece has landed, every fragment is read
        return;
    }

    // =============================== multipliers ======================================================
    __builtin_amdgcn_s_setprio(2);
    constexpr int nRdA = (LA == LAY_K) ? 2 : 4, nRdB = (LB == LAY_K) ? 2 : 4;
    uint32_t rdA[R][nRdA], rdB[R][nRdB];
#pragma unroll
    for (int P = 0; P < R; ++P) {
#pragma unroll
        for (int x = 0; x < nRdA; ++x) {
            rdA[P][x] = ldsBase + (uint32_t)((2 * P) * kHalfBytes) + (LA == LAY_K ? (uint32_t)(8192 * wr) + x_offK(lane, x) : x_offF(lane, 4 * wr + x));
            asm volatile("" : "+v"(rdA[P][x]));
        }
#pragma unroll
        for (int x = 0; x < nRdB; ++x) {
            rdB[P][x] = ldsBase + (uint32_t)((2 * P + 1) * kHalfBytes) + (LB == LAY_K ? (uint32_t)(8192 * wc) + x_offK(lane, x) : x_offF(lane, 4 * wc + x));
            asm volatile("" : "+v"(rdB[P][x]));
        }
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 a[2][4], b[2][4];                       // two register sets: k-step s uses set s

#define CTAMD_W_READ(P, S, Q)                                                                                       \
    {                                                                                                              \
        if constexpr ((Q) < 4) {                                                                                   \
            if constexpr (LB == LAY_K) b[S][Q] = v_read<LAY_K, 2048 * (Q)>(rdB[P][(S) % nRdB]);                    \
            else b[S][Q] = v_read<LAY_F, 8192 * (S)>(rdB[P][(Q) % nRdB]);                                          \
        } else {                                                                                                   \
            if constexpr (LA == LAY_K) a[S][(Q) - 4] = v_read<LAY_K, 2048 * ((Q) - 4)>(rdA[P][(S) % nRdA]);        \
            else a[S][(Q) - 4] = v_read<LAY_F, 8192 * (S)>(rdA[P][((Q) - 4) % nRdA]);                              \
        }                                                                                                          \
    }
#define CTAMD_W_MFMA(S, M) x_mfma<BF>(acc[(M) >> 2][(M) & 3], a[S][(M) >> 2], b[S][(M) & 3]);
    // k-step 0, group Q: one read of k-step 1 (same buffer), two MFMAs;  k-step 1 (behind the barrier): one read of the next tile's
    // k-step 0 (next buffer), two MFMAs
    // (the eight reads of a k-step go out in its FIRST four groups, two per group: at the tile barrier's lgkmcnt(0) the youngest read
    // is then 128 MFMA cycles old instead of 32 — with 512 cycles of MFMAs per K-tile an LDS latency per tile is a quarter of the loop)
#define CTAMD_W_G0(P, Q)                                                                                            \
    if constexpr ((Q) < 4) { CTAMD_W_READ(P, 1, 2 * (Q)) CTAMD_W_READ(P, 1, 2 * (Q) + 1) }                         \
    CTAMD_W_MFMA(0, 2 * (Q)) CTAMD_W_MFMA(0, 2 * (Q) + 1)                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define CTAMD_W_G1(P, Q)                                                                                            \
    if constexpr ((Q) < 4) { CTAMD_W_READ(((P) + 1) % R, 0, 2 * (Q)) CTAMD_W_READ(((P) + 1) % R, 0, 2 * (Q) + 1) } \
    CTAMD_W_MFMA(1, 2 * (Q)) CTAMD_W_MFMA(1, 2 * (Q) + 1)                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define CTAMD_W_TILE(P)                                                                                             \
    CTAMD_W_G0(P, 0) CTAMD_W_G0(P, 1) CTAMD_W_G0(P, 2) CTAMD_W_G0(P, 3)                                            \
    CTAMD_W_G0(P, 4) CTAMD_W_G0(P, 5) CTAMD_W_G0(P, 6) CTAMD_W_G0(P, 7)                                            \
    CTAMD_H_LGKM0();                                                                                               \
    __builtin_amdgcn_s_barrier();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_W_G1(P, 0) CTAMD_W_G1(P, 1) CTAMD_W_G1(P, 2) CTAMD_W_G1(P, 3)                                            \
    CTAMD_W_G1(P, 4) CTAMD_W_G1(P, 5) CTAMD_W_G1(P, 6) CTAMD_W_G1(P, 7)

    __builtin_amdgcn_s_barrier();                 // #0: tile 0 has landed
    __builtin_amdgcn_sched_barrier(0);
    CTAMD_W_READ(0, 0, 0) CTAMD_W_READ(0, 0, 1) CTAMD_W_READ(0, 0, 2) CTAMD_W_READ(0, 0, 3)
    CTAMD_W_READ(0, 0, 4) CTAMD_W_READ(0, 0, 5) CTAMD_W_READ(0, 0, 6) CTAMD_W_READ(0, 0, 7)
    int t = 0;
    if constexpr (R == 4) {
        for (; t + 3 < nTiles; t += 4) { CTAMD_W_TILE(0) CTAMD_W_TILE(1) CTAMD_W_TILE(2) CTAMD_W_TILE(3) }
    } else {
        for (; t + 4 < nTiles; t += 5) { CTAMD_W_TILE(0) CTAMD_W_TILE(1) CTAMD_W_TILE(2) CTAMD_W_TILE(3) CTAMD_W_TILE(4) }
    }
    if (t < nTiles) { CTAMD_W_TILE(0) }
    if (t + 1 < nTiles) { CTAMD_W_TILE(1) }
    if (t + 2 < nTiles) { CTAMD_W_TILE(2) }
    if constexpr (R == 5) { if (t + 3 < nTiles) { CTAMD_W_TILE(3) } }
    CTAMD_H_LGKM0();                              // the (unused) fragments of the tile behind the last one
    x_acc_ready(acc);
    __builtin_amdgcn_s_setprio(0);
    const int laneE = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));

    const uint32_t mW = m0 + 64 * wr, nW = n0 + 64 * wc;      // this wave's quadrant
    GettParams pe;                                // the epilogue's arguments in one burst of scalar loads (gett_h16w4q_kernel)
    h_reload_params(pe);
    if (splitK) {                                 // split-K: fp32 partial tile, row-major [slice][l][m][n]
        const uint32_t Mt = pe.gM.total, Nt = pe.gN.total;
        float* P = pe.partial + ((size_t)slice * pe.gL.total + l) * (size_t)Mt * Nt;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t m = mW + 16 * i + 4 * (laneE >> 4) + r;
                if (m < Mt) {
                    float* row = P + (size_t)m * Nt;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t n = nW + 16 * j + (laneE & 15);
                        if (n < Nt) row[n] = acc[i][j][r];
                    }
                }
            }
        return;
    }
    __builtin_amdgcn_s_barrier();                 // every wave has finished reading the ring, every piece has landed (the movers' last barrier)
    HEpilogue ep;
    ep.init(pe, l, lds, wave);                     // 16 KiB of the (dead) ring per multiplying wave
    if (ep.vecD && ep.beta == 0.f) {
        uint16_t* stage = reinterpret_cast<uint16_t*>(ep.scratch);
        constexpr int kPitch = 72;                // 16-bit elements per image row (gett_h16w4m_kernel's 16-bit epilogue)
        if (ep.flat && mW + 64u <= ep.Mtot && nW + 64u <= ep.Ntot) {      // interior quadrant: pipelined, no bounds tests
            const int64_t sM = pe.gM.stride[1][0];
            x_store_interior16<BF, 4, 4, kPitch>(acc, stage, ep.alpha,
                                                 ep.D + (int64_t)(mW + (uint32_t)(laneE >> 3)) * sM + (int64_t)(nW + 8u * (uint32_t)(laneE & 7)), 8 * sM, laneE);
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4& c = acc[2 * i + a2][j];
                    uint16_t* st = stage + (16 * a2 + 4 * (laneE >> 4)) * kPitch + 16 * j + (laneE & 15);
                    st[0] = h_round16<BF>(ep.alpha * c[0]); st[kPitch] = h_round16<BF>(ep.alpha * c[1]);
                    st[2 * kPitch] = h_round16<BF>(ep.alpha * c[2]); st[3 * kPitch] = h_round16<BF>(ep.alpha * c[3]);
                }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = it * 64 + laneE, row = q >> 3, cc = q & 7;
                const s16x8 v = *reinterpret_cast<const s16x8*>(stage + row * kPitch + 8 * cc);
                const uint32_t m = mW + 32 * i + row, n = nW + 8 * cc;
                if (m < ep.Mtot && n < ep.Ntot) {
                    int64_t offD, offC;
                    ep.offsets(pe, m, n, offD, offC);
                    ep.store16(ep.D + offD, v, n);
                }
            }
        }
    } else {
#pragma unroll
        for (int F = 0; F < 4; ++F)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float* st = ep.scratch + F * 1024 + (16 * (h >> 1) + 4 * (laneE >> 4)) * 32 + 16 * (h & 1) + (laneE & 15);
                const f32x4& c = acc[2 * (F >> 1) + (h >> 1)][2 * (F & 1) + (h & 1)];
                st[0] = ep.alpha * c[0]; st[32] = ep.alpha * c[1]; st[64] = ep.alpha * c[2]; st[96] = ep.alpha * c[3];
            }
        ep.template flush<BF>(pe, mW, 32u, 0u, nW, 0u, 32u, laneE);
    }
}

// =====================================================================================================
// gett_h16w4q_kernel (CUTENSOR_AMD_H16_WAVES=4q): the 64 x 64 x 64 tile for SMALL 16-bit problems (1024^3: 64 tiles of 128 x 128
// leave three quarters of the CUs idle for 16 K-tiles; split-K pays a second launch and a round trip of fp32 partials).
// Same instruction (v_mfma_f32_16x16x32 from inline asm), K odometer, LDS-DMA staging and source-side swizzles; per K-tile an
// operand is 64 rows x 64 k = 8 KiB = two 1-KiB pieces per wave:
//   K-contiguous operand: the first 64 rows of the 128-row half-tile image ([row][64 k], unit u of row r at u ^ ((r >> 1) & 7));
//   free-contiguous operand: image [64 k][64 rows] with 128-byte k-rows, unit p of k-row k holds row-unit p ^ 2 (k & 3): the
//   4 k x 16 rows block a 16-lane group of a transposing read fetches (32 bytes in each of four k-rows) covers all 32 banks.
// 4 waves as 2 x 2, a 32 x 32 quadrant each = 2 x 2 accumulator fragments (16 AGPRs); per K-tile and wave 8 MFMAs, 8 fragment reads
// (tile t + 1 into the other of two register sets while tile t multiplies), 4 LDS-DMA pieces (tile t + 4: ring of four 16-KiB
// K-tiles = 64 KiB, two workgroups per CU), ONE barrier:
//   barrier #(t+1): every wave's pieces of tile t + 1 have landed (vmcnt(8): tiles t + 2, t + 3 stay in flight) and every wave holds
//   tile t's fragments in registers -> buffer t % 4 is refilled with tile t + 4.
// The tile is bound by its staging stream (16 KiB per 128 cycles of MFMA issue), like its 128 x 128 sibling; what it buys is 4 x
// the workgroups.  Epilogue: ONE 64 x 64 16-bit image per workgroup (beta = 0, 16-byte lanes in D: whole 128-byte row segments
// leave), else a 32 x 32 fp32 fragment per wave through HEpilogue::flush.
// =====================================================================================================
constexpr int kQTile = 64;
constexpr int kQBuf = 16384;                      // one K-tile: A 8 KiB + B 8 KiB

template <int LAY>
struct QOperand {
    uint32_t src[2];                              // byte offset of this lane's 16-byte unit, piece i of this wave, relative to `base`
    uint64_t base;
    __device__ __forceinline__ void init(const ModeGroup& gFree, int64_t strideK0, uint32_t row0, int wave, int lane) {
        int64_t off[2];
        int64_t mn = INT64_MAX;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = wave + 4 * i;           // 1-KiB piece 0..7
            if constexpr (LAY == LAY_K) {
                const int r = 8 * c + (lane >> 3), pp = lane & 7;
                const int u = pp ^ ((r >> 1) & 7);
                uint32_t row = row0 + (uint32_t)r;
                if (row >= gFree.total) row = gFree.total - 1;        // clamped rows feed outputs that are never stored
                off[i] = (group_offset<0>(gFree, row) + 8 * u) * 2;
            } else {
                const int kk = 8 * c + (lane >> 3), pp = lane & 7;
                const int u = pp ^ (2 * (kk & 3));
                uint32_t row = row0 + 8u * (uint32_t)u;
                if (row >= gFree.total) row = (gFree.total - 1u) & ~7u;   // the last unit that holds rows of the mode (HOperand::init)
                off[i] = (group_offset<0>(gFree, row) + (int64_t)kk * strideK0) * 2;
            }
            mn = off[i] < mn ? off[i] : mn;
        }
        const int64_t mnW = (int64_t)h_uniform64((uint64_t)h_wave_min(mn));
        base = (uint64_t)mnW;
        src[0] = (uint32_t)(off[0] - mnW);
        src[1] = (uint32_t)(off[1] - mnW);
    }
};

// ragged K / operands without 16-byte lanes (x_rag_mask, x_rag_fix) for QOperand's pieces: K-contiguous — k-unit
// (lane & 7) ^ ((4 wave + (lane >> 4)) & 7) of the lane's row in both pieces; free-contiguous — piece i holds k-row 8 wave + 32 i + (lane >> 3).
// Returns the units (bit i) that hold live data but reach past the end of the tensor: masked here, loaded element-wise by q_rag_fix.
template <int LAY>
__device__ __forceinline__ uint32_t q_rag_mask(uint32_t (&src)[2], int wave, uint32_t kValid, uint32_t limit) {
    const uint32_t laneM = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t strad = 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool out = (LAY == LAY_K) ? (8u * ((laneM & 7u) ^ ((4u * (uint32_t)wave + (laneM >> 4)) & 7u)) >= kValid)
                                        : (8u * (uint32_t)wave + 32u * (uint32_t)i + (laneM >> 3) >= kValid);
        const bool past = !out && src[i] + 16u > limit;
        strad |= past ? (1u << i) : 0u;
        src[i] |= (out || past) ? 0x80000000u : 0u;
    }
    return strad;
}
// sweep-ragged K (x_rag_toggle) for QOperand's pieces: bit 31 of q_rag_mask's `out` lanes flipped
template <int LAY>
__device__ __forceinline__ void q_rag_toggle(uint32_t (&src)[2], int wave, uint32_t kValid) {
    const uint32_t laneM = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool out = (LAY == LAY_K) ? (8u * ((laneM & 7u) ^ ((4u * (uint32_t)wave + (laneM >> 4)) & 7u)) >= kValid)
                                        : (8u * (uint32_t)wave + 32u * (uint32_t)i + (laneM >> 3) >= kValid);
        src[i] ^= out ? 0x80000000u : 0u;
    }
}
// x_rag_fix for the 64 x 64 tile: ldsOp = LDS byte address of the operand's 8-KiB image in the buffer that holds the masked tile
template <int LAY, int SLOTK>
__device__ __forceinline__ void q_rag_fix(const ModeGroup& gFree, const ModeGroup& gK, uint64_t opBase, uint32_t row0, uint32_t kTile0, uint32_t kValid,
                                          uint32_t strad, uint32_t ldsOp, int wave) {
    const uint32_t laneM = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t c = (uint32_t)wave + 4u * (uint32_t)i;
        const uint32_t at = ldsOp + c * 1024u + laneM * 16u;
        const bool past = ((strad >> i) & 1u) != 0u;
        if constexpr (LAY == LAY_K) {
            const uint32_t u = (laneM & 7u) ^ ((4u * (uint32_t)wave + (laneM >> 4)) & 7u);
            const uint32_t v = kValid > 8u * u ? (kValid - 8u * u < 8u ? kValid - 8u * u : 8u) : 0u;
            if (!past && v > 0u && v < 8u) x_zero_tail(at, v);
            if (past) {
                uint32_t row = row0 + 8u * c + (laneM >> 3);
                if (row >= gFree.total) row = gFree.total - 1u;
                const uint64_t addr = opBase + (uint64_t)((group_offset<0>(gFree, row) + group_offset<SLOTK>(gK, kTile0 + 8u * u)) * 2);
                x_patch_unit(at, addr, v);
            }
        } else {
            if (past) {
                const uint32_t kk = 8u * c + (laneM >> 3);
                const uint32_t u = (laneM & 7u) ^ (2u * (kk & 3u));
                uint32_t row = row0 + 8u * u;
                if (row >= gFree.total) row = (gFree.total - 1u) & ~7u;
                const uint32_t nv = gFree.total - row < 8u ? gFree.total - row : 8u;
                const uint64_t addr = opBase + (uint64_t)((group_offset<0>(gFree, row) + group_offset<SLOTK>(gK, kTile0 + kk)) * 2);
                x_patch_unit(at, addr, nv);
            }
        }
    }
}

// byte offset of this lane's 8 bytes of fragment f (rows 16 f) inside the free-contiguous 8-KiB image, first transposing read (k-rows
// 8 g + [0,4) of k-step 0; + 512: k + 4; + 4096: k-step 1)
__device__ __forceinline__ uint32_t q_offF(int lane, int f) {
    const int g = lane >> 4, i = lane & 15;
    const int unit = (2 * f + ((i >> 1) & 1)) ^ (2 * ((i >> 2) & 3));
    return (uint32_t)((8 * g + (i >> 2)) * 128 + (unit << 4) + 8 * (i & 1));
}
template <int LAY, int IMM>
__device__ __forceinline__ s16x8 q_read(uint32_t base) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LAY == LAY_K) {
        return *(VLdsVec8)(uintptr_t)(base + (uint32_t)IMM);
    } else {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VLdsVec4)(uintptr_t)(base + (uint32_t)IMM));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VLdsVec4)(uintptr_t)(base + (uint32_t)IMM + 512u));
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#else
    (void)base; return s16x8{};
#endif
}

// TIMED (measurement, CUTENSOR_AMD_H16_TIMED=1, bf16 mk,kn only): wave 0 records shader cycles at entry / first piece issued / tile 0
// landed / end of the main loop / stores issued and the wall clock at entry / exit (tools/h16_small_timeline.py)
// (An eight-deep ring, 128 KiB and one workgroup per CU, measured the same 460 cycles per K-tile at 1024^3 as this four-deep one and
// a later first tile: a lone workgroup is not latency-bound either.  profiles/r04v_small_timeline.jsonl)
// RAG: ragged K, as in gett_h16w4x_kernel (q_rag_mask): tile t stages tile t + 4 whole, one switch.
template <bool BF, int LA, int LB, bool TIMED = false, bool RAG = false>
__global__ void __launch_bounds__(256, 2) gett_h16w4q_kernel(const GettParams p) {
    constexpr int R = 4;
    __shared__ __attribute__((aligned(16))) char lds[R * kQBuf];
    unsigned long long qs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (TIMED) { qs[0] = __builtin_readcyclecounter(); qs[5] = wall_clock64(); }
    prefetch_kernarg<(int)sizeof(GettParams)>();
    // the arguments the setup reads, in ONE burst of scalar loads (through `p` they arrive one dependent round at a time)
    GettParams ps;
    h_reload_params(ps);
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    uint32_t id = xcd_remap(blockIdx.x, ps.nBlocks);
    // the launch's tiles: rectangle 1 (tilesM x tilesN from (mOrg, nOrg)), then rectangle 2 (the other edge strip of a strip plan)
    const uint32_t tiles1 = ps.tilesM * ps.tilesN;
    const uint32_t tilesMN = tiles1 + ps.tilesM2 * ps.tilesN2;
    const uint32_t tilesAll = tilesMN * ps.gL.total;
    const uint32_t slice = id / tilesAll;
    id -= slice * tilesAll;
    const uint32_t l = id / tilesMN;
    id -= l * tilesMN;
    const bool second = id >= tiles1;
    id -= second ? tiles1 : 0u;
    const uint32_t rTilesM = second ? ps.tilesM2 : ps.tilesM, rTilesN = second ? ps.tilesN2 : ps.tilesN;
    const uint32_t perGroup = 8u * rTilesN;
    const uint32_t grp = id / perGroup, inGrp = id - grp * perGroup;
    const uint32_t first = grp * 8u;
    const uint32_t gsz = (rTilesM - first < 8u) ? (rTilesM - first) : 8u;
    const uint32_t mt = first + inGrp % gsz, nt = inGrp / gsz;
    const uint32_t m0 = (second ? ps.mOrg2 : ps.mOrg) + mt * kQTile, n0 = (second ? ps.nOrg2 : ps.nOrg) + nt * kQTile;
    const bool sweep = RAG && (ps.ragged & 2u) != 0u;         // sweep-ragged K: gett_h16w4x_kernel
    const uint32_t kTilesAll = sweep ? (ps.ragged >> 2) : (ps.gK.total + (RAG ? (uint32_t)kHBK - 1u : 0u)) / kHBK, tilesPerSlice = ps.kPerSlice / kHBK;
    const uint32_t tile0 = slice * tilesPerSlice;
    const int nTiles = (int)((tile0 + tilesPerSlice <= kTilesAll) ? tilesPerSlice : (kTilesAll - tile0));
    if constexpr (TIMED) { asm volatile("" :: "s"(nTiles), "s"(m0), "s"(n0)); qs[7] = __builtin_readcyclecounter(); }   // arguments fetched, tile located

    QOperand<LA> oa;
    QOperand<LB> ob;
    oa.init(ps.gM, ps.gK.stride[0][0], m0, wave, lane);
    ob.init(ps.gN, ps.gK.stride[1][0], n0, wave, lane);
    const uint64_t bA = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(ps.A) + group_offset<0>(ps.gL, l)) + oa.base);
    const uint64_t bB = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(ps.B) + group_offset<1>(ps.gL, l)) + ob.base);
    VOdometer odo;
    if (sweep) odo.init_tiles(ps.gK, tile0, (uint32_t)nTiles, bA, bB);
    else odo.template init<RAG>(ps.gK, tile0 * kHBK, (uint32_t)nTiles, bA, bB);
    // ragged K / operands without 16-byte lanes: index (among this workgroup's K-tiles) of the tile that is staged masked — the last K-tile
    // of the last slice — and how many k of the contracted range it holds (q_rag_mask, q_rag_fix)
    const int maskAt = (RAG && !sweep && tile0 + (uint32_t)nTiles == kTilesAll) ? nTiles - 1 : 0x7fffffff;
    const uint32_t kValid = VOdometer::sgpr(sweep ? ps.gK.div[0].d - (odo.n0 - 1u) * (uint32_t)kHBK
                                                  : ((ps.gK.total % kHBK) != 0u ? ps.gK.total % kHBK : (uint32_t)kHBK));
    uint32_t stradA = 0u, stradB = 0u;
    bool maskOn = false;                          // sweep-ragged K: the masked lanes are out of range right now
    // called with the descriptor bases ON tile IDX (sweep: tiles past the slice's last one are re-staged copies of it — the mask stays)
#define CTAMD_Q_RAGMASK(IDX)                                                                                        \
    if constexpr (RAG) {                                                                                           \
        if ((IDX) == maskAt) {                                                                                     \
            stradA = q_rag_mask<LA>(oa.src, wave, kValid, x_rag_limit(p.endA, odo.addrA));                         \
            stradB = q_rag_mask<LB>(ob.src, wave, kValid, x_rag_limit(p.endB, odo.addrB));                         \
        }                                                                                                          \
        if (sweep && (IDX) < nTiles && odo.on_sweep_end() != maskOn) {                                             \
            q_rag_toggle<LA>(oa.src, wave, kValid);                                                                \
            q_rag_toggle<LB>(ob.src, wave, kValid);                                                                \
            maskOn = !maskOn;                                                                                      \
        }                                                                                                          \
    }
    // tile IDX (the masked one) has landed in buffer PB, behind a workgroup barrier (gett_h16w4x_kernel, CTAMD_X_RAGFIX)
#define CTAMD_Q_RAGFIX(IDX, PB)                                                                                     \
    if constexpr (RAG) {                                                                                           \
        if ((IDX) == maskAt) {                                                                                     \
            const bool fixA = (LA == LAY_K) ? (kValid & 7u) != 0u : ((p.gM.total & 7u) != 0u && m0 + (uint32_t)kQTile >= p.gM.total);   \
            const bool fixB = (LB == LAY_K) ? (kValid & 7u) != 0u : ((p.gN.total & 7u) != 0u && n0 + (uint32_t)kQTile >= p.gN.total);   \
            if (fixA || fixB) {                                                                                    \
                const uint32_t kT0 = (kTilesAll - 1u) * (uint32_t)kHBK;                                            \
                if (fixA) q_rag_fix<LA, 0>(p.gM, p.gK, (uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)), m0, kT0, kValid, stradA, \
                                           ldsBase + (uint32_t)((PB) * kQBuf), wave);                              \
                if (fixB) q_rag_fix<LB, 1>(p.gN, p.gK, (uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)), n0, kT0, kValid, stradB, \
                                           ldsBase + (uint32_t)((PB) * kQBuf + 8192), wave);                       \
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
                __builtin_amdgcn_s_barrier();                                                                      \
            }                                                                                                      \
        }                                                                                                          \
    }

    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t waveLds = VOdometer::sgpr(ldsBase + (uint32_t)wave * 1024u);
    // fragment-read address registers: this wave's fragments are f = 2 wr + i (A) / 2 wc + j (B).  K-contiguous operand: one register
    // per k-step (fragment 2048 i and buffer in the immediate); free-contiguous: one per fragment (k-step 4096 s and buffer in the
    // immediate)
    uint32_t rdA[2], rdB[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        rdA[x] = ldsBase + (LA == LAY_K ? (uint32_t)(4096 * wr) + x_offK(lane, x) : q_offF(lane, 2 * wr + x));
        rdB[x] = ldsBase + 8192u + (LB == LAY_K ? (uint32_t)(4096 * wc) + x_offK(lane, x) : q_offF(lane, 2 * wc + x));
        asm volatile("" : "+v"(rdA[x]));
        asm volatile("" : "+v"(rdB[x]));
    }

    // piece N = 0..3 of the K-tile the odometer describes into buffer P: operand N >> 1 (A, B), piece N & 1 of this wave
#define CTAMD_Q_DMA(P, N, PAD)                                                                                      \
    {                                                                                                              \
        constexpr uint32_t imm_ = (uint32_t)((P) * kQBuf + ((N) >> 1) * 8192 + ((N) & 1) * 4096);                  \
        if constexpr (((N) >> 1) == 0) v_dma16<imm_, PAD>(v_rsrc<RAG>(odo.addrA), oa.src[(N) & 1], waveLds);       \
        else v_dma16<imm_, PAD>(v_rsrc<RAG>(odo.addrB), ob.src[(N) & 1], waveLds);                                 \
    }
#define CTAMD_Q_DMA4(P, PAD) CTAMD_Q_DMA(P, 0, PAD) CTAMD_Q_DMA(P, 1, PAD) CTAMD_Q_DMA(P, 2, PAD) CTAMD_Q_DMA(P, 3, PAD)
#define CTAMD_Q_NEXT() { odo.advance_a(); odo.advance_b(); odo.advance_event(p.gK); }

    // ---- prologue: K-tiles 0 .. R - 1; the odometer stays on tile R - 1 ----------------------------------------------------
    if constexpr (TIMED) qs[1] = __builtin_readcyclecounter();
    CTAMD_Q_RAGMASK(0) CTAMD_Q_DMA4(0, true)
    CTAMD_Q_NEXT() CTAMD_Q_RAGMASK(1) CTAMD_Q_DMA4(1, true)
    CTAMD_Q_NEXT() CTAMD_Q_RAGMASK(2) CTAMD_Q_DMA4(2, true)
    CTAMD_Q_NEXT() CTAMD_Q_RAGMASK(3) CTAMD_Q_DMA4(3, true)
    CTAMD_H_VMCNT(4 * (R - 1));                   // this wave's pieces of tile 0
    __builtin_amdgcn_s_barrier();
    CTAMD_Q_RAGFIX(0, 0)
    if constexpr (TIMED) qs[2] = __builtin_readcyclecounter();

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 a[2][2][2], b[2][2][2];                 // [register set][k-step][fragment]

    // the eight fragments of the K-tile in buffer P into register set S
#define CTAMD_Q_RD(P, S, KS, F)                                                                                     \
    if constexpr (LA == LAY_K) a[S][KS][F] = q_read<LAY_K, (P) * kQBuf + 2048 * (F)>(rdA[KS]);                     \
    else a[S][KS][F] = q_read<LAY_F, (P) * kQBuf + 4096 * (KS)>(rdA[F]);                                           \
    if constexpr (LB == LAY_K) b[S][KS][F] = q_read<LAY_K, (P) * kQBuf + 2048 * (F)>(rdB[KS]);                     \
    else b[S][KS][F] = q_read<LAY_F, (P) * kQBuf + 4096 * (KS)>(rdB[F]);
#define CTAMD_Q_READ8(P, S) CTAMD_Q_RD(P, S, 0, 0) CTAMD_Q_RD(P, S, 0, 1) CTAMD_Q_RD(P, S, 1, 0) CTAMD_Q_RD(P, S, 1, 1)
#define CTAMD_Q_MFMA2(S, KS, I) x_mfma<BF>(acc[I][0], a[S][KS][I], b[S][KS][0]); x_mfma<BF>(acc[I][1], a[S][KS][I], b[S][KS][1]);
    // tile t in buffer P / register set P & 1
#define CTAMD_Q_TILE(P)                                                                                             \
    CTAMD_H_VMCNT(4 * (R - 2));       /* tile t + 1 has landed; the pieces of tiles t + 2 .. t + R - 1 stay in flight */ \
    __builtin_amdgcn_s_barrier();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_Q_RAGFIX(t + (P) + 1, ((P) + 1) % R)                                                                     \
    CTAMD_Q_READ8(((P) + 1) % R, ((P) + 1) & 1)                                                                    \
    CTAMD_Q_NEXT()                                                                                                 \
    CTAMD_Q_RAGMASK(t + (P) + 4)                                                                                   \
    CTAMD_Q_DMA(P, 0, false) CTAMD_Q_MFMA2((P) & 1, 0, 0)                                                          \
    CTAMD_Q_DMA(P, 1, false) CTAMD_Q_MFMA2((P) & 1, 0, 1)                                                          \
    CTAMD_Q_DMA(P, 2, false) CTAMD_Q_MFMA2((P) & 1, 1, 0)                                                          \
    CTAMD_Q_DMA(P, 3, false) CTAMD_Q_MFMA2((P) & 1, 1, 1)                                                          \
    CTAMD_H_LGKM0();                                                                                               \
    __builtin_amdgcn_sched_barrier(0);

    CTAMD_Q_READ8(0, 0)
    CTAMD_H_LGKM0();
    int t = 0;
    for (; t + 3 < nTiles; t += 4) { CTAMD_Q_TILE(0) CTAMD_Q_TILE(1) CTAMD_Q_TILE(2) CTAMD_Q_TILE(3) }
    if (t < nTiles) { CTAMD_Q_TILE(0) }
    if (t + 1 < nTiles) { CTAMD_Q_TILE(1) }
    if (t + 2 < nTiles) { CTAMD_Q_TILE(2) }
    CTAMD_H_VMCNT(0);                             // the re-staged tail: no LDS-DMA may outlive the workgroup
    x_acc_ready(acc);
    if constexpr (TIMED) qs[3] = __builtin_readcyclecounter();
    const int laneE = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));

    const uint32_t mW = m0 + 32 * wr, nW = n0 + 32 * wc;      // this wave's quadrant
    // a fresh copy of the arguments for the epilogue, loaded in ONE burst (only the fields used): read through `p` the compiler fetches
    // them one dependent s_load round at a time (~20 rounds, most of a 64 x 64 tile's epilogue: tools/h16_small_timeline.py)
    GettParams pe;
    h_reload_params(pe);
    // accumulator fragment (i, j): element r of laneE = row 16 i + 4 (laneE >> 4) + r, column 16 j + (laneE & 15)
    if (pe.partial != nullptr) {                   // split-K: fp32 partial tile, row-major [slice][l][m][n]
        const uint32_t Mt = pe.gM.total, Nt = pe.gN.total;
        float* P = pe.partial + ((size_t)slice * pe.gL.total + l) * (size_t)Mt * Nt;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t m = mW + 16 * i + 4 * (laneE >> 4) + r;
                if (m < Mt) {
                    float* row = P + (size_t)m * Nt;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const uint32_t n = nW + 16 * j + (laneE & 15);
                        if (n < Nt) row[n] = acc[i][j][r];
                    }
                }
            }
        return;
    }
    __syncthreads();                              // every wave has finished reading the operand ring, no piece is in flight
    HEpilogue ep;
    ep.init(pe, l, lds + 16384, wave, 4096);       // fp32 path: 4 KiB per wave behind the 16-bit image's 9 KiB
    if (ep.vecD && ep.beta == 0.f) {
        // ONE 16-bit image of the workgroup's 64 x 64 tile (144-byte rows: the 2-byte writes of a 16-lane group and the 16-byte reads
        // of a row both spread over the banks); out as 16-byte reads + nontemporal stores of whole 128-byte row segments
        uint16_t* stage = reinterpret_cast<uint16_t*>(lds);
        constexpr int kPitch = 72;                // 16-bit elements per image row
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4& c = acc[i][j];
                uint16_t* st = stage + (32 * wr + 16 * i + 4 * (laneE >> 4)) * kPitch + 32 * wc + 16 * j + (laneE & 15);
                st[0] = h_round16<BF>(ep.alpha * c[0]); st[kPitch] = h_round16<BF>(ep.alpha * c[1]);
                st[2 * kPitch] = h_round16<BF>(ep.alpha * c[2]); st[3 * kPitch] = h_round16<BF>(ep.alpha * c[3]);
            }
        __syncthreads();
        const int tidE = wave * 64 + laneE;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * 256 + tidE, row = q >> 3, cc = q & 7;
            const s16x8 v = *reinterpret_cast<const s16x8*>(stage + row * kPitch + 8 * cc);
            const uint32_t m = m0 + (uint32_t)row, n = n0 + 8u * (uint32_t)cc;
            if (m < ep.Mtot && n < ep.Ntot) {
                int64_t offD, offC;
                ep.offsets(pe, m, n, offD, offC);
                ep.store16(ep.D + offD, v, n);
            }
        }
    } else {
        // the wave's 32 x 32 quadrant as one fp32 fragment
#pragma unroll
        for (int h = 0; h < 4; ++h) {             // 16 x 16 quarter (h >> 1, h & 1)
            float* st = ep.scratch + (16 * (h >> 1) + 4 * (laneE >> 4)) * 32 + 16 * (h & 1) + (laneE & 15);
            const f32x4& c = acc[h >> 1][h & 1];
            st[0] = ep.alpha * c[0]; st[32] = ep.alpha * c[1]; st[64] = ep.alpha * c[2]; st[96] = ep.alpha * c[3];
        }
        ep.template flush<BF, 0, 1>(pe, mW, 0u, 0u, nW, 0u, 0u, laneE);
    }
    if constexpr (TIMED) {
        if (p.timing != nullptr && wave == 0 && laneE == 0) {
            qs[4] = __builtin_readcyclecounter();                 // the stores are issued, not waited for
            qs[6] = wall_clock64();
#pragma unroll
            for (int i = 0; i < 8; ++i) p.timing[64 + 8 * (size_t)blockIdx.x + i] = qs[i];
        }
    }
}

template <bool BF, int LA, int LB>
static hipError_t launch_h16w4q(const GettParams& p, hipStream_t stream) {
#if defined(CTAMD_RESEARCH_KERNELS)
    if constexpr (BF && LA == LAY_K && LB == LAY_F) {
        static const bool timed = [] { const char* e = getenv("CUTENSOR_AMD_H16_TIMED"); return e && e[0] == '1'; }();
        if (timed) { hipLaunchKernelGGL((gett_h16w4q_kernel<BF, LA, LB, true>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
    }
#endif
    if (p.gK.total % (uint32_t)kHBK != 0u || (p.ragged & 1u) != 0u) {   // ragged K (one contracted mode) or a unit that can straddle the tensor's end (pick_h16_choice): the masked last K-tile
        hipLaunchKernelGGL((gett_h16w4q_kernel<BF, LA, LB, false, true>), dim3(p.nBlocks), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((gett_h16w4q_kernel<BF, LA, LB>), dim3(p.nBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <bool BF, int LA, int LB>
static hipError_t launch_h16w8m(const GettParams& p, hipStream_t stream) {
#if defined(CTAMD_RESEARCH_KERNELS)
    static const bool ring4 = [] { const char* e = getenv("CUTENSOR_AMD_H16_RING"); return e && e[0] == '4'; }();
#else
    static const bool ring4 = false;
#endif   // measurement: the four-deep ring
    if (ring4) hipLaunchKernelGGL((gett_h16w8m_kernel<BF, LA, LB, 4>), dim3(p.nBlocks), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((gett_h16w8m_kernel<BF, LA, LB, 5>), dim3(p.nBlocks), dim3(512), 0, stream, p);
    return hipGetLastError();
}

template <bool BF, int LA, int LB, int R>
static hipError_t launch_h16w4m(const GettParams& p, hipStream_t stream) {
    if (p.gK.total % (uint32_t)kHBK != 0u || (p.ragged & 1u) != 0u) {   // ragged K (one contracted mode) or a unit that can straddle the tensor's end (pick_h16_choice): the masked last K-tile
        hipLaunchKernelGGL((gett_h16w4m_kernel<BF, LA, LB, R, true>), dim3(p.nBlocks), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((gett_h16w4m_kernel<BF, LA, LB, R>), dim3(p.nBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

#if defined(CTAMD_RESEARCH_KERNELS)
template <bool BF, int LA, int LB>
static hipError_t launch_h16w4v(const GettParams& p, hipStream_t stream) {
    hipLaunchKernelGGL((gett_h16w4v_kernel<BF, LA, LB>), dim3(p.nBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// bf16 entries first, then fp16, each in the order (layA, layB) = (K,K) (K,F) (F,K) (F,F) — the order of gett_h16.hip's table
#define CTAMD_H16W4V_ENTRY(bf, la, lb) \
    {kHTile, kHTile, kHBK, 2, 2, 1, la, lb, 256, 6, 1, 0, &launch_h16w4v<bf, la, lb>, 0},
#else
static hipError_t launch_h16v_not_built(const GettParams&, hipStream_t) { return hipErrorNotSupported; }
#define CTAMD_H16W4V_ENTRY(bf, la, lb) {kHTile, kHTile, kHBK, 2, 2, 1, la, lb, 256, 6, 1, 2, &launch_h16v_not_built, 0},
#endif
#define CTAMD_H16W4X_ENTRY(bf, la, lb) \
    {kHTile, kHTile, kHBK, 2, 2, 1, la, lb, 256, 7, 1, 0, &launch_h16w4x<bf, la, lb>, 0},
#define CTAMD_H16W4M_ENTRY(bf, la, lb) \
    {kMTile, kMTile, kHBK, 2, 2, 1, la, lb, 256, 8, 1, 0, &launch_h16w4m<bf, la, lb, 2>, 0},
#define CTAMD_H16W4M4_ENTRY(bf, la, lb) \
    {kMTile, kMTile, kHBK, 2, 2, 1, la, lb, 256, 9, 1, 0, &launch_h16w4m<bf, la, lb, 4>, 0},
#define CTAMD_H16W8M_ENTRY(bf, la, lb) \
    {kMTile, kMTile, kHBK, 2, 2, 1, la, lb, 512, 10, 1, 0, &launch_h16w8m<bf, la, lb>, 0},
#define CTAMD_H16W4Q_ENTRY(bf, la, lb) \
    {kQTile, kQTile, kHBK, 2, 2, 1, la, lb, 256, 11, 1, 0, &launch_h16w4q<bf, la, lb>, 0},
static const GettKernelInfo g_h16v_table[] = {
    CTAMD_H16W4V_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W4V_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W4V_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W4V_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W4V_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W4V_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W4V_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W4V_ENTRY(false, LAY_F, LAY_F)
    // entries 8..15 of this table (48..55 of the 16-bit family): the 16x16x32 form
    CTAMD_H16W4X_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W4X_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W4X_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W4X_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W4X_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W4X_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W4X_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W4X_ENTRY(false, LAY_F, LAY_F)
    // entries 16..23 (56..63 of the family): the 128 x 128 mid-size sibling, two workgroups per CU
    CTAMD_H16W4M_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W4M_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W4M_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W4M_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W4M_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W4M_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W4M_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W4M_ENTRY(false, LAY_F, LAY_F)
    // entries 24..31 (64..71 of the family): the same with a four-deep K-tile ring, one workgroup per CU
    CTAMD_H16W4M4_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W4M4_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W4M4_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W4M4_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W4M4_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W4M4_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W4M4_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W4M4_ENTRY(false, LAY_F, LAY_F)
    // entries 32..39 (72..79 of the family): 128 x 128, four multiplying + four data-moving waves, four-deep ring
    CTAMD_H16W8M_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W8M_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W8M_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W8M_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W8M_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W8M_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W8M_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W8M_ENTRY(false, LAY_F, LAY_F)
    // entries 40..47 (80..87 of the family): 64 x 64, four waves, four-deep ring of 16-KiB K-tiles, two workgroups per CU
    CTAMD_H16W4Q_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W4Q_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W4Q_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W4Q_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W4Q_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W4Q_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W4Q_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W4Q_ENTRY(false, LAY_F, LAY_F)};

const GettKernelInfo* gett_h16v_kernels(int* count) {
    *count = (int)(sizeof(g_h16v_table) / sizeof(g_h16v_table[0]));
    return g_h16v_table;
}

}  // namespace ctamd
