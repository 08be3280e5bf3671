// gett_h16p.hip — gett_h16w4p_kernel (round 5): the PERSISTENT form of gett_h16w4x_kernel (gett_h16v.hip).
//
// Same tile (256 x 256 x 64), four waves (one per SIMD, a 128 x 128 quadrant = 8 x 8 accumulator fragments of
// v_mfma_f32_16x16x32_{bf16,f16} from inline asm), LDS images, source-side swizzles, LDS-DMA staging, K odometer and main loop.
// What changes is everything OUTSIDE the main loop — the 6.6 % of a workgroup's 313k cycles at 8192^3 (prologue 7.8k, epilogue
// 12-13.7k; profiles/r05a_w4x_barrier_decomposition.jsonl) and more than half of them at 8192^2 x 512.  Measured result (NOTES.md, round
// 5a): 8192^2 x 512 +6.8 %, x 1024 +3.3 %, x 2048 +2.5 %, x 4096 +1.8 %, 8192^3 +0.8 % on U(-1,1) data (the chip is power-limited
// there: cycles that used to idle carry MFMAs and the clock goes down by about as much), +10 % on zeros; one-round shapes -1 %.
//   * one workgroup per CU walks the tiles (grid = CUs rounded down to a multiple of 8; tile ids in the XCD-grouped order of the
//     one-tile kernel: workgroup w takes the ids w, w + grid, w + 2 grid, ... of xcd_remap's sequence, so an XCD keeps its 8 x 4
//     block of concurrent tiles);
//   * interior tiles STREAM into each other (the tile in flight and the next one inside D of a problem with one M and one N mode and
//     16-byte lanes in D — batch modes included since round 6: the hand-over computes the next tile's batch offset as it does its row
//     offsets; an even K-tile count): the per-lane staging
//     offsets do not depend on the tile — only the descriptor bases move — so K-tile nTiles - 2 hands the odometer to the next tile
//     (CTAMD_P_SWITCH, in a copy of the body pair of its own) and the LDS-DMA pieces the last two K-tile bodies issue anyway fetch the
//     next tile's K-tiles 0 and 1; the next tile starts on a peeled pair without vmcnt(0).  No setup, no staging latency between tiles;
//   * the streamed tile's epilogue does not use the operand ring (it is being refilled): a wave owns images in the 32 KiB of LDS beyond
//     the 128-KiB ring — 160 KiB in all — and works in passes of 16 rows;
//   * the image is TRANSPOSED: an accumulator fragment holds, per lane, four consecutive ROWS of one column, so after two packed
//     conversions (v_cvt_pk_bf16_f32) the lane's four values are 8 contiguous bytes of a column-major image [128 columns][16 rows]
//     — ONE ds_write_b64 per fragment instead of four 2-byte writes — and ds_read_b64_tr_b16, the transposing read the main loop
//     uses for free-contiguous operands, brings four consecutive columns of a row to every lane.  They are copied
//     into a second, row-major image and leave as four 256-byte row segments per lane group; storing straight from the transposing
//     reads gives 64-byte segments and is 5 % slower (profiles/r05b, r05c).  Image row of column c at 32 R(c) bytes,
//     R(c) = c ^ 4 ((c >> 3) & 1); the 8-byte slot of rows 4 s .. 4 s + 3 inside it at s ^ ((R >> 2) & 3): the ds_write_b64 of a
//     16-lane group and the transposing read of a 32-lane half both touch every bank once (gett_h16p_layout.h, replayed on the CPU:
//     tests/test_gen_layout_cpu.py::test_persistent_kernel_epilogue_image_replay);
//   * every global access of the epilogue goes through an address_space(1) pointer: a pending FLAT store makes each counted
//     lgkmcnt wait of the next tile's first K-tiles a full drain.
//   * beta != 0 (round 6): C joins the accumulators in fp32 before the ONE rounding and the tile still streams — the pass's 16 rows of
//     C are loaded the way D is stored, written row-major into the row image while it is idle and brought into the accumulator layout
//     by the transposing read run the other way round (four ROWS of one column per lane; p_c_write_off / p_c_read_off, replayed on the
//     CPU with the other images).  Same bits as the one-tile kernel's fp32-image epilogue (tests/test_gpu_h16p.py); on one box 8192^3
//     beta = 0.5 1.48-1.53 PFLOP/s against 1.42-1.45 on the one-tile kernel, 8192^2 x 2048 1.30-1.32 against 1.04-1.07, x 1024
//     1.03-1.06 against 0.76-0.77 (profiles/r06r_h16p_beta_vs_one_tile_ab.jsonl, r06s_h16p_beta_depth_ab.jsonl).
// Every other tile (edges, strided D, a C without the 16-byte lanes of D, batch modes, split-K partials) takes the epilogues of
// gett_h16w4x_kernel, in the ring, and is set up and staged behind them — slower than the one-tile kernel, which is why the planner
// offers this kernel only to problems whose interior tiles can stream and cutensorContract launches the one-tile twin for beta != 0 with
// such a C (plan_contraction.cpp, api.cpp).  Roofline and algorithmic bytes as in gett_h16.hip (MFMA bf16; 2 M N K flop).
#include <type_traits>

#include "gett_h16x_common.h"
#include "gett_h16p_layout.h"

namespace ctamd {

constexpr int kPRingBytes  = 8 * kHalfBytes;   // two K-tiles of 64 KiB
constexpr int kPImageBytes = kPImgBytes;       // one pass of a wave: 128 columns x 16 rows x 2 B (gett_h16p_layout.h)
// Height (in tiles along M) of the column-major tile groups the ids walk: with 32 consecutive ids per XCD and round, 8 gives every XCD an
// 8 x 4 block of concurrent tiles = 12 operand panels per K-step.  -DCTAMD_P_XCD_GROUP=4 / 16 are the measurement builds of round 6
// (4 x 8: the same 12 panels with the roles of A and B exchanged; 16 x 2: 18 panels; profiles/r06zt_*).
#ifndef CTAMD_P_XCD_GROUP
#define CTAMD_P_XCD_GROUP 8
#endif
constexpr uint32_t kPGroup = CTAMD_P_XCD_GROUP;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// four consecutive rows of one column (an accumulator fragment's four registers, times alpha) -> four 16-bit values
template <bool BF>
__device__ __forceinline__ s16x4 p_round4(const f32x4& c, float alpha) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (BF) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 lo = __builtin_convertvector(f32x2{alpha * c[0], alpha * c[1]}, bf16x2);
        const bf16x2 hi = __builtin_convertvector(f32x2{alpha * c[2], alpha * c[3]}, bf16x2);
        const uint32_t l = __builtin_bit_cast(uint32_t, lo), h = __builtin_bit_cast(uint32_t, hi);
        return __builtin_bit_cast(s16x4, (unsigned long long)l | ((unsigned long long)h << 32));
    } else {
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        const f16x2 lo = __builtin_convertvector(f32x2{alpha * c[0], alpha * c[1]}, f16x2);
        const f16x2 hi = __builtin_convertvector(f32x2{alpha * c[2], alpha * c[3]}, f16x2);
        const uint32_t l = __builtin_bit_cast(uint32_t, lo), h = __builtin_bit_cast(uint32_t, hi);
        return __builtin_bit_cast(s16x4, (unsigned long long)l | ((unsigned long long)h << 32));
    }
#else
    (void)c; (void)alpha; return s16x4{};
#endif
}

// ... the same with beta * C added before the ONE rounding (cv: the four 16-bit values of C at the fragment's positions): h_round16_with_c,
// the arithmetic of every HEpilogue path (gett_h16_common.h), so both kernels of a plan give the same bits
template <bool BF>
__device__ __forceinline__ s16x4 p_round4c(const f32x4& c, float alpha, float beta, const s16x4& cv) {
    return s16x4{(short)h_round16_with_c<BF>(alpha * c[0], beta, (uint16_t)cv[0]), (short)h_round16_with_c<BF>(alpha * c[1], beta, (uint16_t)cv[1]),
                 (short)h_round16_with_c<BF>(alpha * c[2], beta, (uint16_t)cv[2]), (short)h_round16_with_c<BF>(alpha * c[3], beta, (uint16_t)cv[3])};
}

// (Round 5 compared three ways out of the transposed image with a TIMED instantiation of this kernel — a second, row-major image and
// stores of 4 rows x 256 bytes; straight from the transposing reads, stores of 16 rows x 64 bytes, nontemporal or plain: 5 % slower,
// profiles/r05b, r05c_h16p_epilogue_variants.jsonl.  The row image stayed; the variants and the in-kernel time stamps, which had read
// zeros since the tiles stream into each other, were removed in round 6.)
// VOdometer::advance_event with the K mode table read through a LAUNDERED argument pointer inside the rare branch: handed `p.gK`, the
// compiler hoists the table's ~30 scalar loads out of the tile loop (they are loop-invariant) and keeps them alive — spilled — across
// the main loop, whose bodies then carry v_readlane / v_writelane between their MFMAs.
__device__ __forceinline__ void p_advance_event(VOdometer& odo) {
#if defined(__HIP_DEVICE_COMPILE__)
    odo.untilEvent -= 1u;
    if (__builtin_expect(odo.untilEvent == 0u, 0)) {
        auto kp = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        typedef const __attribute__((address_space(4))) GettParams* PArg;
        const ModeGroup gK = ((PArg)kp)->gK;
        if (odo.carryPending != 0u) {                      // the advance just made wrapped digit 1: bases from the full index
            odo.hi += 1u;
            const uint32_t k = odo.hi * odo.e1 * gK.div[0].d;
            if (k < gK.total) {
                odo.addrA = h_uniform64(odo.baseA + (uint64_t)(group_offset<0>(gK, k) * 2));
                odo.addrB = h_uniform64(odo.baseB + (uint64_t)(group_offset<1>(gK, k) * 2));
            }
        }
        odo.next_segment(odo.carryLen);
    }
#else
    (void)odo;
#endif
}

template <bool BF, int LA, int LB>
__global__ void __launch_bounds__(256, 1) gett_h16w4p_kernel(const GettParams p) {
    __shared__ __attribute__((aligned(16))) char lds[kPRingBytes + 8 * kPImageBytes];     // 160 KiB: the ring + two pass images per wave
    prefetch_kernarg<(int)sizeof(GettParams)>();
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // staggered start (GettParams::mOrg2, a field only the strip kernel uses otherwise; set by launch_h16w4p): with a short contracted
    // range a tile is mostly epilogue, and 256 workgroups that start together store together and idle the memory side together — the
    // workgroups of an XCD start (id / 8) % 4 quarter-phases apart (units of 1024 cycles per phase step)
    if (p.mOrg2 != 0u) {
        const uint32_t steps = ((blockIdx.x >> 3) & 3u) * p.mOrg2;
        for (uint32_t i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(16);
    }

    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t waveLds = VOdometer::sgpr(ldsBase + (uint32_t)wave * 1024u);
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

    // ---- the tile in flight ------------------------------------------------------------------------------------------------
    HOperand<LA, 4, false, 1> oa;
    HOperand<LB, 4, false, 1> ob;
    VOdometer odo;
    uint32_t m0 = 0, n0 = 0, slice = 0, l = 0;
    int nTiles = 0;
    // ---- streaming across tiles (see CTAMD_P_SWITCH) ---------------------------------------------------------------------------
    uint64_t relA = 0, relB = 0;                  // tile-invariant parts of the staging bases: base = rel + 2 * row0 * stride (interior tiles, flat M / N)
    bool curOK = false;                           // the tile in flight lies inside D of a problem whose interior tiles take the fast epilogue
    bool streamedOut = false;                     // the NEXT tile's first two K-tiles were issued by this tile's last two K-tile bodies
    bool streamedIn = false;                      // ... and this tile arrived that way
    int nTilesNext = 0;
    // tile of virtual workgroup id VB_ (the one-tile kernel's blockIdx.x): coordinates, staging tables, K odometer.  The arguments come
    // from a FRESH copy of the argument block and the lane index from the hardware, behind an opaque asm: nothing this block needs
    // may stay live across the main loop (kept in registers "because it is used again" it would be ~150 SGPRs of mode tables and a
    // dozen lane-derived VGPRs — spilled, and reloaded by v_readlane between the MFMAs)
#define CTAMD_P_SETUP(VB_)                                                                                          \
    {                                                                                                              \
        GettParams ps_;                                                                                            \
        h_reload_params(ps_);                                                                                      \
        int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));                       \
        asm volatile("" : "+v"(lane_));                                                                            \
        const uint32_t tilesMN_ = ps_.tilesM * ps_.tilesN;                                                         \
        const uint32_t tilesAll_ = tilesMN_ * ps_.gL.total;                                                        \
        const uint32_t kTilesAll_ = ps_.gK.total / kHBK, tilesPerSlice_ = ps_.kPerSlice / kHBK;                    \
        uint32_t id_ = xcd_remap((VB_), ps_.nBlocks);                                                              \
        slice = VOdometer::sgpr(id_ / tilesAll_);                                                                  \
        id_ -= slice * tilesAll_;                                                                                  \
        l = VOdometer::sgpr(id_ / tilesMN_);                                                                       \
        id_ -= l * tilesMN_;                                                                                       \
        const uint32_t perGroup_ = kPGroup * ps_.tilesN;                                                                \
        const uint32_t grp_ = id_ / perGroup_, inGrp_ = id_ - grp_ * perGroup_;                                    \
        const uint32_t first_ = grp_ * kPGroup;                                                                         \
        const uint32_t gsz_ = (ps_.tilesM - first_ < kPGroup) ? (ps_.tilesM - first_) : kPGroup;                             \
        m0 = VOdometer::sgpr((first_ + inGrp_ % gsz_) * kHTile);                                                   \
        n0 = VOdometer::sgpr((inGrp_ / gsz_) * kHTile);                                                            \
        const uint32_t tile0_ = slice * tilesPerSlice_;                                                            \
        nTiles = (int)VOdometer::sgpr((tile0_ + tilesPerSlice_ <= kTilesAll_) ? tilesPerSlice_ : (kTilesAll_ - tile0_)); \
        oa.init(ps_.gM, ps_.gK.stride[0][0], m0, wave, lane_);                                                     \
        ob.init(ps_.gN, ps_.gK.stride[1][0], n0, wave, lane_);                                                     \
        const uint64_t bA_ = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(ps_.A) + group_offset<0>(ps_.gL, l)) + oa.base); \
        const uint64_t bB_ = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(ps_.B) + group_offset<1>(ps_.gL, l)) + ob.base); \
        odo.init(ps_.gK, tile0_ * kHBK, (uint32_t)nTiles, bA_, bB_);                                               \
        nTiles += (int)VOdometer::sgpr((ps_.partial == nullptr && (nTiles & 1) != 0) ? 1u : 0u);   /* odd count: a zero K-tile is appended (VOdometer::recs) */ \
        HEpilogue e_;                                                                                              \
        e_.init(ps_, l, lds, wave);                                                                                \
        curOK = VOdometer::sgpr((e_.vecD && (e_.beta == 0.f || e_.vecC) && e_.flat && ps_.partial == nullptr &&                        \
                                 m0 + (uint32_t)kHTile <= ps_.gM.total && n0 + (uint32_t)kHTile <= ps_.gN.total) ? 1u : 0u) != 0u;  \
        relA = h_uniform64(oa.base - (uint64_t)m0 * (uint64_t)ps_.gM.stride[0][0] * 2ull);                         \
        relB = h_uniform64(ob.base - (uint64_t)n0 * (uint64_t)ps_.gN.stride[0][0] * 2ull);                         \
    }
// The tile AFTER the one in flight, staged by the main loop itself: called in k-step 0 of K-tile nTiles - 2 (instead of the odometer's
// event check — the odometer has nothing left to do there: its last real advance moved it to K-tile nTiles - 1), it re-initialises the
// odometer for the next tile of this workgroup's walk, so that the LDS-DMA pieces the last two K-tile bodies issue anyway ("K-tile t + 2":
// a re-staged, never-read tile in gett_h16w4x_kernel) fetch the next tile's K-tiles 0 and 1 — no extra issue slot, no setup or staging
// between the tiles.  What makes it cheap: for tiles that lie inside a flat problem the per-lane staging offsets (HOperand::src, 16
// VGPRs) do not depend on the tile — only the descriptor bases move (rel + 2 * row0 * stride) — so the "setup" is scalar arithmetic.
// Needs: this tile and the next one interior (this one takes the epilogue that stays out of the ring), an even K-tile count (the next
// tile starts in ring buffer 0); a streamed-in tile's first pair of bodies is a copy without the first barrier's vmcnt(0), and with two K-tiles per tile that pair is the hand-over as well (a third copy, round 6).  Otherwise the odometer's event check runs and the next tile is
// staged the slow way, under / after the epilogue.
#define CTAMD_P_SWITCH()                                                                                            \
    {                                                                                                              \
        GettParams pn_;                                                                                            \
        h_reload_params(pn_);                                                                                      \
        const uint32_t nv_ = vb + gridX;                                                                           \
        bool ok_ = false;                                                                                          \
        if (VOdometer::sgpr(nv_ < pn_.nBlocks ? 1u : 0u) != 0u) {                                                  \
            const uint32_t tilesMN_ = pn_.tilesM * pn_.tilesN;                                                     \
            const uint32_t tilesAll_ = tilesMN_ * pn_.gL.total;                                                    \
            const uint32_t kTilesAll_ = pn_.gK.total / kHBK, tilesPerSlice_ = pn_.kPerSlice / kHBK;                \
            uint32_t id_ = xcd_remap(nv_, pn_.nBlocks);                                                            \
            const uint32_t slice_ = VOdometer::sgpr(id_ / tilesAll_);                                              \
            id_ -= slice_ * tilesAll_;                                                                             \
            const uint32_t l_ = VOdometer::sgpr(id_ / tilesMN_);                                                   \
            id_ -= l_ * tilesMN_;                                                                                  \
            const uint32_t perGroup_ = kPGroup * pn_.tilesN;                                                            \
            const uint32_t grp_ = id_ / perGroup_, inGrp_ = id_ - grp_ * perGroup_;                                \
            const uint32_t first_ = grp_ * kPGroup;                                                                     \
            const uint32_t gsz_ = (pn_.tilesM - first_ < kPGroup) ? (pn_.tilesM - first_) : kPGroup;                         \
            const uint32_t m0n_ = VOdometer::sgpr((first_ + inGrp_ % gsz_) * kHTile);                              \
            const uint32_t n0n_ = VOdometer::sgpr((inGrp_ / gsz_) * kHTile);                                       \
            const uint32_t tile0_ = slice_ * tilesPerSlice_;                                                       \
            const uint32_t nt_ = VOdometer::sgpr((tile0_ + tilesPerSlice_ <= kTilesAll_) ? tilesPerSlice_ : (kTilesAll_ - tile0_)); \
            if (VOdometer::sgpr((m0n_ + (uint32_t)kHTile <= pn_.gM.total && n0n_ + (uint32_t)kHTile <= pn_.gN.total && nt_ >= 1u) ? 1u : 0u) != 0u) { \
                const uint64_t bA_ = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(pn_.A) + group_offset<0>(pn_.gL, l_)) + relA + \
                                                 (uint64_t)m0n_ * (uint64_t)pn_.gM.stride[0][0] * 2ull);           \
                const uint64_t bB_ = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(pn_.B) + group_offset<1>(pn_.gL, l_)) + relB + \
                                                 (uint64_t)n0n_ * (uint64_t)pn_.gN.stride[0][0] * 2ull);           \
                odo.init(pn_.gK, tile0_ * kHBK, nt_, bA_, bB_);                                                    \
                nTilesNext = (int)(nt_ + (nt_ & 1u));                                                              \
                ok_ = true;                                                                                        \
            }                                                                                                      \
        }                                                                                                          \
        streamedOut = ok_;                                                                                         \
        if (!ok_) p_advance_event(odo);                                                                            \
    }
#define CTAMD_P_DMA(P, N, PAD)                                                                                      \
    {                                                                                                              \
        constexpr int q_ = (N) >> 2, i_ = (N) & 3;                                                                 \
        constexpr uint32_t imm_ = (uint32_t)(((P) * 4 + q_) * kHalfBytes + i_ * 4096);                             \
        if constexpr (q_ < 2) v_dma16<imm_, PAD>(v_rsrc_n(odo.addrA, odo.recs), oa.src[q_][i_], waveLds);          \
        else v_dma16<imm_, PAD>(v_rsrc_n(odo.addrB, odo.recs), ob.src[q_ - 2][i_], waveLds);                       \
    }
#define CTAMD_P_DMA8(P, N0, PAD)                                                                                    \
    CTAMD_P_DMA(P, (N0) + 0, PAD) CTAMD_P_DMA(P, (N0) + 1, PAD) CTAMD_P_DMA(P, (N0) + 2, PAD) CTAMD_P_DMA(P, (N0) + 3, PAD) \
    CTAMD_P_DMA(P, (N0) + 4, PAD) CTAMD_P_DMA(P, (N0) + 5, PAD) CTAMD_P_DMA(P, (N0) + 6, PAD) CTAMD_P_DMA(P, (N0) + 7, PAD)
    // K-tiles 0 and 1 of the tile just set up; the odometer stays on tile 1 (k-step 0 of tile t moves it to tile t + 2)
#define CTAMD_P_ISSUE2()                                                                                            \
    {                                                                                                              \
        CTAMD_P_DMA8(0, 0, true) CTAMD_P_DMA8(0, 8, true)                                                          \
        odo.advance_a(); odo.advance_b(); p_advance_event(odo);                                                    \
        CTAMD_P_DMA8(1, 0, true) CTAMD_P_DMA8(1, 8, true)                                                          \
    }

    f32x4 acc[8][8];
    s16x8 a[2][8], b[2][8];                       // two register sets: k-step s uses set s
#define CTAMD_P_READ(P, S, Q)                                                                                       \
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
#define CTAMD_P_MFMA(S, M) x_mfma<BF>(acc[(M) >> 3][(M) & 7], a[S][(M) >> 3], b[S][(M) & 7]);
#else
#define CTAMD_P_MFMA(S, M) x_mfma<BF>(acc[(M) >> 3][(((M) >> 3) & 1) ? ((M) & 7) : 7 - ((M) & 7)], a[S][(M) >> 3], b[S][(((M) >> 3) & 1) ? ((M) & 7) : 7 - ((M) & 7)]);
#endif
    // k-step 0, group Q: one read of k-step 1 (same buffer) and four MFMAs; three of the groups carry the odometer
#define CTAMD_P_G0(P, Q, SW)                                                                                        \
    CTAMD_P_READ(P, 1, Q)                                                                                          \
    CTAMD_P_MFMA(0, 4 * (Q)) CTAMD_P_MFMA(0, 4 * (Q) + 1)                                                          \
    if constexpr ((Q) == 2) odo.advance_a();                                                                       \
    if constexpr ((Q) == 5) odo.advance_b();                                                                       \
    if constexpr ((Q) == 8) {                                                                                      \
        if constexpr (SW) CTAMD_P_SWITCH() else p_advance_event(odo);                                              \
    }                                                                                                              \
    CTAMD_P_MFMA(0, 4 * (Q) + 2) CTAMD_P_MFMA(0, 4 * (Q) + 3)                                                      \
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1 (behind the barrier), group Q: one read of the next tile's k-step 0 (other buffer), one piece of tile t + 2 into
    // this buffer, four MFMAs
// (group 0: the read goes BEHIND the first MFMA pair — inside this kernel's outer tile loop the compiler's wait-count bookkeeping
// puts an s_waitcnt lgkmcnt(0) in front of the first MFMA after the tile barrier instead of gett_h16w4x_kernel's lgkmcnt(9); with the
// read in front of it that wait cost the read's whole latency, +100 cycles per K-tile: profiles/r05d_h16p_timeline.jsonl)
#define CTAMD_P_G1(P, Q)                                                                                            \
    if constexpr ((Q) != 0) CTAMD_P_READ((P) ^ 1, 0, Q)                                                            \
    CTAMD_P_MFMA(1, 4 * (Q)) CTAMD_P_MFMA(1, 4 * (Q) + 1)                                                          \
    if constexpr ((Q) == 0) CTAMD_P_READ((P) ^ 1, 0, Q)                                                            \
    CTAMD_P_DMA(P, Q, false)                                                                                       \
    CTAMD_P_MFMA(1, 4 * (Q) + 2) CTAMD_P_MFMA(1, 4 * (Q) + 3)                                                      \
    __builtin_amdgcn_sched_barrier(0);
// VM = false: the tile barrier without its vmcnt(0) — K-tile 1 of a streamed-in tile is known to have landed (every wave waited for
// it inside the previous tile's epilogue, before its first store), and a vmcnt(0) here would wait for that epilogue's stores
// SW = true: the body that hands the odometer over to the next tile (CTAMD_P_SWITCH in place of the odometer's event check) — a copy
// of its own, executed once per tile behind the main pair loop: a test inside the hot bodies cost a compare, a branch and seven scalar
// copies (the merge of two odometer states) per K-tile pair
#define CTAMD_P_TILE_(P, VM, SW)                                                                                    \
    CTAMD_P_G0(P, 0, SW) CTAMD_P_G0(P, 1, SW) CTAMD_P_G0(P, 2, SW) CTAMD_P_G0(P, 3, SW) CTAMD_P_G0(P, 4, SW) CTAMD_P_G0(P, 5, SW)      \
    CTAMD_P_G0(P, 6, SW) CTAMD_P_G0(P, 7, SW) CTAMD_P_G0(P, 8, SW) CTAMD_P_G0(P, 9, SW) CTAMD_P_G0(P, 10, SW) CTAMD_P_G0(P, 11, SW)    \
    CTAMD_P_G0(P, 12, SW) CTAMD_P_G0(P, 13, SW) CTAMD_P_G0(P, 14, SW) CTAMD_P_G0(P, 15, SW)                        \
    CTAMD_H_LGKM0();                                                                                               \
    if constexpr (VM) CTAMD_H_VMCNT(0);                                                                            \
    __builtin_amdgcn_s_barrier();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_P_G1(P, 0) CTAMD_P_G1(P, 1) CTAMD_P_G1(P, 2) CTAMD_P_G1(P, 3) CTAMD_P_G1(P, 4) CTAMD_P_G1(P, 5)          \
    CTAMD_P_G1(P, 6) CTAMD_P_G1(P, 7) CTAMD_P_G1(P, 8) CTAMD_P_G1(P, 9) CTAMD_P_G1(P, 10) CTAMD_P_G1(P, 11)        \
    CTAMD_P_G1(P, 12) CTAMD_P_G1(P, 13) CTAMD_P_G1(P, 14) CTAMD_P_G1(P, 15)
#define CTAMD_P_TILE(P) CTAMD_P_TILE_(P, true, false)

    // The grid size NOW, through an opaque asm: left to the compiler, its scalar load is hoisted above the main loop and waited for
    // behind it — and while a scalar load may be pending every LDS wait in the loop has to be lgkmcnt(0) (scalar loads return out of
    // order): the fragment reads issued right behind the tile barrier were waited for before the first MFMA of the k-step, +100
    // cycles per K-tile (profiles/r05d_h16p_timeline.jsonl: 2368 against gett_h16w4x_kernel's 2271).
    uint32_t gridX = gridDim.x;
    asm volatile("" : "+s"(gridX));
    uint32_t vb = blockIdx.x;                     // virtual workgroup id of the tile in flight
    bool staged = false;                          // its first two K-tiles are already on their way (issued under the previous epilogue)
    // (CTAMD_PROBE_ONEPASS / _NOPARTIAL / _NOGENERAL: compile-time probes for reading the ISA — one tile per workgroup, no split-K
    // epilogue, no general epilogue — that isolate what each part of the tile loop does to register allocation; never defined by the
    // Makefile, wrong results when defined)
#if defined(CTAMD_PROBE_ONEPASS)
    for (int once_ = 0; once_ < 1; ++once_) {
#else
    for (;;) {
#endif
        if (!staged) {
            CTAMD_P_SETUP(vb)
            CTAMD_P_ISSUE2()
        }
        const int curTiles = nTiles;
        // the K-tile body (even index, ring buffer 0) that hands the odometer over to the next tile: the last but one, when this tile
        // may stream its successor in
        const int switchAt = (curOK && (curTiles & 1) == 0 && curTiles >= 2) ? curTiles - 2 : -1;
        streamedOut = false;
        // K-tile 0 has landed: loads complete in issue order, so "at most 16 memory operations outstanding" leaves at most the 16
        // pieces of K-tile 1 (and, from the second tile on, waits out the previous epilogue's stores, which were issued later).  A
        // streamed-in tile: every wave has waited for both K-tiles inside the previous epilogue.
        if (!streamedIn) CTAMD_H_VMCNT(16);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        CTAMD_P_READ(0, 0, 0) CTAMD_P_READ(0, 0, 1) CTAMD_P_READ(0, 0, 2) CTAMD_P_READ(0, 0, 3)
        CTAMD_P_READ(0, 0, 4) CTAMD_P_READ(0, 0, 5) CTAMD_P_READ(0, 0, 6) CTAMD_P_READ(0, 0, 7)
        CTAMD_P_READ(0, 0, 8) CTAMD_P_READ(0, 0, 9) CTAMD_P_READ(0, 0, 10) CTAMD_P_READ(0, 0, 11)
        CTAMD_P_READ(0, 0, 12) CTAMD_P_READ(0, 0, 13) CTAMD_P_READ(0, 0, 14) CTAMD_P_READ(0, 0, 15)
        int t = 0;
        // a streamed-in tile starts on a peeled pair (no vmcnt(0) at its first tile barrier).  With TWO K-tiles per tile (round 6: K = 128 —
        // attention scores, 'ik,kj->ij' with a short k) that pair is also the one that hands the odometer over: a third copy of the pair
        const bool onlyPair = streamedIn && curTiles == 2;
        if (onlyPair) { CTAMD_P_TILE_(0, false, true) CTAMD_P_TILE(1) t = 2; }
        else if (streamedIn) { CTAMD_P_TILE_(0, false, false) CTAMD_P_TILE(1) t = 2; }
        streamedIn = false;
        const int pairEnd = switchAt >= 0 ? switchAt : curTiles;
        for (; t + 1 < pairEnd; t += 2) { CTAMD_P_TILE(0) CTAMD_P_TILE(1) }
        if (switchAt >= 0 && !onlyPair) { CTAMD_P_TILE_(0, true, true) CTAMD_P_TILE(1) t += 2; }      // K-tiles nTiles - 2 (the hand-over) and nTiles - 1
        if (t < curTiles) { CTAMD_P_TILE(0) }
        if (!streamedOut) CTAMD_H_VMCNT(0);       // the re-staged tail: no LDS-DMA may be in flight when the ring is staged again
        x_acc_ready(acc);
        // the lane index again, from the hardware: nothing lane-derived stays live across the main loop for the epilogue's sake
        const int laneE = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const uint32_t nextVb = vb + gridX;
        GettParams pe;                            // the epilogue's arguments in one burst of scalar loads
        h_reload_params(pe);
        // the tile's coordinates AGAIN, from its id: only `vb` and the K-tile count stay live across the main loop (six more scalar
        // registers there were four v_writelane per K-tile in the loop — spill traffic between the MFMAs, +80 cycles per K-tile)
        uint32_t tM0, tN0, curL, curSlice;
        {
            const uint32_t tilesMN_ = pe.tilesM * pe.tilesN, tilesAll_ = tilesMN_ * pe.gL.total;
            uint32_t id_ = xcd_remap(vb, pe.nBlocks);
            curSlice = VOdometer::sgpr(id_ / tilesAll_);
            id_ -= curSlice * tilesAll_;
            curL = VOdometer::sgpr(id_ / tilesMN_);
            id_ -= curL * tilesMN_;
            const uint32_t perGroup_ = kPGroup * pe.tilesN;
            const uint32_t grp_ = id_ / perGroup_, inGrp_ = id_ - grp_ * perGroup_;
            const uint32_t first_ = grp_ * kPGroup;
            const uint32_t gsz_ = (pe.tilesM - first_ < kPGroup) ? (pe.tilesM - first_) : kPGroup;
            tM0 = VOdometer::sgpr((first_ + inGrp_ % gsz_) * kHTile);
            tN0 = VOdometer::sgpr((inGrp_ / gsz_) * kHTile);
        }
        const uint32_t mW = tM0 + 128 * wr, nW = tN0 + 128 * wc;   // this wave's quadrant
        // (values read through the laundered argument pointer count as divergent for the compiler: what steers control flow is made
        // wave-uniform again explicitly, or the K loop's scalar state would be given vector registers)
        const bool more = VOdometer::sgpr(nextVb < pe.nBlocks ? 1u : 0u) != 0u;
        // accumulator fragment (i, j): element r of laneE = row 16 i + 4 (laneE >> 4) + r, column 16 j + (laneE & 15)
#if !defined(CTAMD_PROBE_NOPARTIAL)
        if (VOdometer::sgpr(pe.partial != nullptr ? 1u : 0u) != 0u) {   // split-K: fp32 partial tile, row-major [slice][l][m][n]; no LDS involved
            const uint32_t Mt = pe.gM.total, Nt = pe.gN.total;
            float* P = pe.partial + ((size_t)curSlice * pe.gL.total + curL) * (size_t)Mt * Nt;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t m = mW + 16 * i + 4 * (laneE >> 4) + r;
                    if (m < Mt) {
                        typedef float __attribute__((address_space(1))) * PGlbF;     // global, not flat, stores (gett_h16_common.h, HGlbS8)
                        PGlbF row = (PGlbF)(uintptr_t)(P + (size_t)m * Nt);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const uint32_t n = nW + 16 * j + (laneE & 15);
                            if (n < Nt) row[n] = acc[i][j][r];
                        }
                    }
                }
            if (!more) break;
            __syncthreads();                      // every wave has finished reading the operand ring
            vb = nextVb;
            staged = false;
            continue;
        }
#endif
        __syncthreads();                          // every wave has finished reading the operand ring
        HEpilogue ep;
        ep.init(pe, curL, lds, wave);
        // workgroup-uniform: the whole tile inside D, one M and one N mode, 16-byte lanes, nothing to add
        const bool fast = streamedOut || VOdometer::sgpr((ep.vecD && (ep.beta == 0.f || ep.vecC) && ep.flat && tM0 + (uint32_t)kHTile <= ep.Mtot && tN0 + (uint32_t)kHTile <= ep.Ntot) ? 1u : 0u) != 0u;
        if (fast) {
            const int64_t sM = pe.gM.stride[1][0];
            const float alpha = ep.alpha;
            if (more && !streamedOut) {           // not streamed in by the main loop: the next tile's first two K-tiles are staged now and arrive under this epilogue
                vb = nextVb;
                CTAMD_P_SETUP(vb)
                CTAMD_P_ISSUE2()
            }
            // image offsets: gett_h16p_layout.h (replayed on the CPU by tests/harness/h16p_layout_harness.cpp)
            char* const img = lds + kPRingBytes + wave * (2 * kPImageBytes);
            char* const wPtr = img + p_img_write_off(laneE, 0);                 // fragment j: + 512 j
            const char* const rPtr0 = img + p_img_read_off(laneE, 0, 0);        // iteration it: + 1024 it
            const char* const rPtr1 = img + p_img_read_off(laneE, 0, 1);
            typedef s16x8 __attribute__((address_space(1))) * PGlb8;            // D is device memory: global, not flat, stores
            typedef s16x4 __attribute__((address_space(3))) * PLds4;
#define CTAMD_P_TR(DST, BUF, IT)                                                                                    \
            {                                                                                                      \
                const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((PLds4)(uintptr_t)(uint32_t)(uintptr_t)(rPtr0 + (BUF) * kPImageBytes + 1024 * (IT))); \
                const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((PLds4)(uintptr_t)(uint32_t)(uintptr_t)(rPtr1 + (BUF) * kPImageBytes + 1024 * (IT))); \
                DST = s16x8{lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]};                         \
            }
            {
                // Two images per wave: T (transposed, [128 columns][16 rows]) and R (row-major, [16 rows][256 bytes], 16-byte units
                // XOR-swizzled by the row).  Pass I, software-pipelined by one pass (the LDS executes a wave's operations in order, so
                // single images suffice): the chunks of pass I - 1 (in t[], from the transposing reads at the end of that pass) are
                // parked in R and fetched back as 4 rows x 16 chunks (v[]); the eight fragments of pass I are converted into T; v[]
                // leaves as four stores of 4 rows x 256 bytes, one per two fragments; the transposing reads of pass I refill t[].
                char* const rowImg = img + kPImageBytes;
                uint16_t* dst2 = ep.D + (int64_t)(mW + (uint32_t)(laneE >> 4)) * sM + (int64_t)(nW + 8u * (uint32_t)(laneE & 15));
                const int64_t step4 = 4 * sM;
                s16x8 t[4], v[4];
#define CTAMD_P_FRAG(I, J)                                                                                          \
                *reinterpret_cast<s16x4*>(wPtr + 512 * (J)) = p_round4<BF>(acc[(I) < 8 ? (I) : 0][J], alpha);
#define CTAMD_P_PARK(IT)  *reinterpret_cast<s16x8*>(rowImg + p_row_write_off(laneE, IT)) = t[IT];
#define CTAMD_P_FETCH(IT) v[IT] = *reinterpret_cast<const s16x8*>(rowImg + p_row_read_off(laneE, IT));
#define CTAMD_P_STORE(IT) { __builtin_nontemporal_store(v[IT], (PGlb8)(uintptr_t)dst2); dst2 += step4; }
#define CTAMD_P_PASS(I)                                                                                             \
                {                                                                                                  \
                    if constexpr ((I) < 8) { CTAMD_P_FRAG(I, 0) CTAMD_P_FRAG(I, 1) }                               \
                    __builtin_amdgcn_sched_barrier(0);                                                             \
                    if constexpr ((I) > 0) {      /* the transposing reads of pass I - 1 have had two fragments' time to return */ \
                        CTAMD_P_PARK(0) CTAMD_P_PARK(1) CTAMD_P_PARK(2) CTAMD_P_PARK(3)                            \
                        CTAMD_P_FETCH(0) CTAMD_P_FETCH(1) CTAMD_P_FETCH(2) CTAMD_P_FETCH(3)                        \
                    }                                                                                              \
                    __builtin_amdgcn_sched_barrier(0);                                                             \
                    if constexpr ((I) < 8) { CTAMD_P_FRAG(I, 2) CTAMD_P_FRAG(I, 3) }                               \
                    __builtin_amdgcn_sched_barrier(0);                                                             \
                    if constexpr ((I) > 0) { CTAMD_P_STORE(0) }                                                    \
                    if constexpr ((I) < 8) { CTAMD_P_FRAG(I, 4) CTAMD_P_FRAG(I, 5) }                               \
                    __builtin_amdgcn_sched_barrier(0);                                                             \
                    if constexpr ((I) > 0) { CTAMD_P_STORE(1) CTAMD_P_STORE(2) }                                   \
                    if constexpr ((I) < 8) { CTAMD_P_FRAG(I, 6) CTAMD_P_FRAG(I, 7) }                               \
                    __builtin_amdgcn_sched_barrier(0);                                                             \
                    if constexpr ((I) > 0) { CTAMD_P_STORE(3) }                                                    \
                    if constexpr ((I) < 8) { CTAMD_P_TR(t[0], 0, 0) CTAMD_P_TR(t[1], 0, 1) CTAMD_P_TR(t[2], 0, 2) CTAMD_P_TR(t[3], 0, 3) } \
                    __builtin_amdgcn_sched_barrier(0);                                                             \
                }
                if (VOdometer::sgpr(ep.beta != 0.f ? 1u : 0u) != 0u) {
                    // beta != 0 (round 6): C joins the accumulators in fp32, BEFORE the one rounding.  The pass's 16 rows of C are loaded
                    // the way D is stored (cg: 4 rows x 256 bytes per instruction, two passes ahead), written row-major into the row image
                    // while it is idle (after the fetch of pass I - 1) and brought into the accumulator layout by the transposing read run
                    // the other way round (cf: fragment j = four rows of one column per lane, read back once the previous pass's fragments are out; gett_h16p_layout.h, p_c_*): 4 loads, 4
                    // ds_write_b128 and 8 ds_read_b64_tr_b16 per pass on top of the beta == 0 pipeline, no second rounding, no fp32 image.
                    const float beta = ep.beta;
                    const int64_t sMc = pe.cStrideM[0];
                    const uint16_t* src2 = ep.C + (int64_t)(mW + (uint32_t)(laneE >> 4)) * sMc + (int64_t)(nW + 8u * (uint32_t)(laneE & 15));
                    const int64_t cstep4 = 4 * sMc;
                    typedef const s16x8 __attribute__((address_space(1))) * PGlbC8;
                    // kPCDepth: register sets of C chunks (16 VGPRs each) = passes of C requested ahead.  Memory operations complete in order,
                    // so a request issued behind the stores of an earlier pass returns once those are acknowledged — but three sets (every
                    // request but the last two in front of the first store) measure the same as two (profiles/r06s_h16p_beta_depth_ab.jsonl:
                    // 8192^3 1.48-1.53 against 1.50-1.52 PFLOP/s, 8192^2 x 1024 1.03-1.05 both), four and more spill vector registers
#if defined(CTAMD_P_C_DEPTH)
                    constexpr int kPCDepth = CTAMD_P_C_DEPTH;
#else
                    constexpr int kPCDepth = 2;
#endif
                    s16x8 cg[kPCDepth][4];
                    s16x4 cf[8];
                    uint32_t cRd[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) cRd[j] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(rowImg + p_c_read_off(laneE, j));
#define CTAMD_P_CG(I)                                                                                               \
                    {                                                                                              \
                        cg[(I) % kPCDepth][0] = *(PGlbC8)(uintptr_t)src2; src2 += cstep4;                                  \
                        cg[(I) % kPCDepth][1] = *(PGlbC8)(uintptr_t)src2; src2 += cstep4;                                  \
                        cg[(I) % kPCDepth][2] = *(PGlbC8)(uintptr_t)src2; src2 += cstep4;                                  \
                        cg[(I) % kPCDepth][3] = *(PGlbC8)(uintptr_t)src2; src2 += cstep4;                                  \
                    }
#define CTAMD_P_CW(I)                                                                                               \
                    {                                                                                              \
                        *reinterpret_cast<s16x8*>(rowImg + p_c_write_off(laneE, 0)) = cg[(I) % kPCDepth][0];               \
                        *reinterpret_cast<s16x8*>(rowImg + p_c_write_off(laneE, 1)) = cg[(I) % kPCDepth][1];               \
                        *reinterpret_cast<s16x8*>(rowImg + p_c_write_off(laneE, 2)) = cg[(I) % kPCDepth][2];               \
                        *reinterpret_cast<s16x8*>(rowImg + p_c_write_off(laneE, 3)) = cg[(I) % kPCDepth][3];               \
                    }
#define CTAMD_P_CR1(I, J) cf[J] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((PLds4)(uintptr_t)cRd[J]);
#define CTAMD_P_CR(I) CTAMD_P_CR1(I, 0) CTAMD_P_CR1(I, 1) CTAMD_P_CR1(I, 2) CTAMD_P_CR1(I, 3) CTAMD_P_CR1(I, 4) CTAMD_P_CR1(I, 5) CTAMD_P_CR1(I, 6) CTAMD_P_CR1(I, 7)
#define CTAMD_P_FRAGC(I, J)                                                                                         \
                    *reinterpret_cast<s16x4*>(wPtr + 512 * (J)) = p_round4c<BF>(acc[(I) < 8 ? (I) : 0][J], alpha, beta, cf[J]);
#define CTAMD_P_PASSC(I)                                                                                            \
                    {                                                                                              \
                        if constexpr ((I) < 8) { CTAMD_P_FRAGC(I, 0) CTAMD_P_FRAGC(I, 1) }                         \
                        __builtin_amdgcn_sched_barrier(0);                                                         \
                        if constexpr ((I) > 0) {                                                                   \
                            CTAMD_P_PARK(0) CTAMD_P_PARK(1) CTAMD_P_PARK(2) CTAMD_P_PARK(3)                        \
                            CTAMD_P_FETCH(0) CTAMD_P_FETCH(1) CTAMD_P_FETCH(2) CTAMD_P_FETCH(3)                    \
                        }                                                                                          \
                        __builtin_amdgcn_sched_barrier(0);                                                         \
                        if constexpr ((I) + 1 < 8) { CTAMD_P_CW((I) + 1) }                        /* the row image is idle: C of the next pass goes into it */ \
                        __builtin_amdgcn_sched_barrier(0);                                                         \
                        if constexpr (kPCDepth < 8 && (I) + 1 + kPCDepth < 8) { CTAMD_P_CG((I) + 1 + kPCDepth) }   /* (its registers were just written out) */ \
                        if constexpr ((I) < 8) { CTAMD_P_FRAGC(I, 2) CTAMD_P_FRAGC(I, 3) }                         \
                        __builtin_amdgcn_sched_barrier(0);                                                         \
                        if constexpr ((I) > 0) { CTAMD_P_STORE(0) }                                                \
                        if constexpr ((I) < 8) { CTAMD_P_FRAGC(I, 4) CTAMD_P_FRAGC(I, 5) }                         \
                        __builtin_amdgcn_sched_barrier(0);                                                         \
                        if constexpr ((I) > 0) { CTAMD_P_STORE(1) CTAMD_P_STORE(2) }                               \
                        if constexpr ((I) < 8) { CTAMD_P_FRAGC(I, 6) CTAMD_P_FRAGC(I, 7) }                         \
                        __builtin_amdgcn_sched_barrier(0);                                                         \
                        if constexpr ((I) + 1 < 8) { CTAMD_P_CR((I) + 1) }                        /* ... and out again, now that this pass's fragments are done with cf */ \
                        if constexpr ((I) > 0) { CTAMD_P_STORE(3) }                                                \
                        if constexpr ((I) < 8) { CTAMD_P_TR(t[0], 0, 0) CTAMD_P_TR(t[1], 0, 1) CTAMD_P_TR(t[2], 0, 2) CTAMD_P_TR(t[3], 0, 3) } \
                        __builtin_amdgcn_sched_barrier(0);                                                         \
                    }
                    // all of C is requested up front (pass 0 goes through the image at once); streamed: the vmcnt(0) that waits for the next
                    // tile's first K-tiles also waits for it, before any store is in flight
                    CTAMD_P_CG(0) CTAMD_P_CG(1)
                    if constexpr (kPCDepth > 2) { CTAMD_P_CG(2) }
                    if constexpr (kPCDepth > 3) { CTAMD_P_CG(3) }
                    if constexpr (kPCDepth > 4) { CTAMD_P_CG(4) }
                    if constexpr (kPCDepth > 5) { CTAMD_P_CG(5) }
                    if constexpr (kPCDepth > 6) { CTAMD_P_CG(6) }
                    if constexpr (kPCDepth > 7) { CTAMD_P_CG(7) }
                    __builtin_amdgcn_sched_barrier(0);
                    CTAMD_P_CW(0) CTAMD_P_CR(0)
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (kPCDepth < 8) { CTAMD_P_CG(kPCDepth) }
                    CTAMD_P_PASSC(0)
                    if (streamedOut) CTAMD_H_VMCNT(0);
                    CTAMD_P_PASSC(1) CTAMD_P_PASSC(2) CTAMD_P_PASSC(3) CTAMD_P_PASSC(4)
                    CTAMD_P_PASSC(5) CTAMD_P_PASSC(6) CTAMD_P_PASSC(7) CTAMD_P_PASSC(8)
#undef CTAMD_P_PASSC
#undef CTAMD_P_FRAGC
#undef CTAMD_P_CR
#undef CTAMD_P_CR1
#undef CTAMD_P_CW
#undef CTAMD_P_CG
                } else {
                CTAMD_P_PASS(0)
                // streamed: the next tile's K-tiles 0 and 1 (issued by the last two K-tile bodies) must have landed before the next main
                // loop reads them — waited for HERE, while no store of this epilogue is in flight yet (vmcnt counts stores too)
                if (streamedOut) CTAMD_H_VMCNT(0);
                CTAMD_P_PASS(1) CTAMD_P_PASS(2) CTAMD_P_PASS(3) CTAMD_P_PASS(4)
                CTAMD_P_PASS(5) CTAMD_P_PASS(6) CTAMD_P_PASS(7) CTAMD_P_PASS(8)
                }
#undef CTAMD_P_PASS
#undef CTAMD_P_STORE
#undef CTAMD_P_FETCH
#undef CTAMD_P_PARK
#undef CTAMD_P_FRAG
            }
#undef CTAMD_P_TR
            if (!more) break;
            if (streamedOut) {                    // the odometer already is the next tile's; its staging tables are this tile's
                vb = nextVb;
                nTiles = nTilesNext;
                curOK = true;                     // (CTAMD_P_SWITCH checked that it lies inside D)
            }
            staged = true;
            streamedIn = streamedOut;
            continue;
        }
        // ---- every other tile: the epilogues of gett_h16w4x_kernel, in the (dead) ring ---------------------------------------------
#if !defined(CTAMD_PROBE_NOGENERAL)
        if (ep.vecD && ep.beta == 0.f) {
            // beta == 0 and 16-byte lanes in D: a pass of 32 rows x 128 columns is a 16-bit image of 272-byte rows, rounded once
            uint16_t* stage = reinterpret_cast<uint16_t*>(ep.scratch);
            constexpr int kPitch = 136;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f32x4& c = acc[2 * i + a2][j];
                        uint16_t* st = stage + (16 * a2 + 4 * (laneE >> 4)) * kPitch + 16 * j + (laneE & 15);
                        st[0] = h_round16<BF>(ep.alpha * c[0]); st[kPitch] = h_round16<BF>(ep.alpha * c[1]);
                        st[2 * kPitch] = h_round16<BF>(ep.alpha * c[2]); st[3 * kPitch] = h_round16<BF>(ep.alpha * c[3]);
                    }
#pragma unroll 4
                for (int it = 0; it < 8; ++it) {
                    const int q = it * 64 + laneE, row = q >> 4, cc = q & 15;
                    const s16x8 v = *reinterpret_cast<const s16x8*>(stage + row * kPitch + 8 * cc);
                    const uint32_t m = mW + 32 * i + row, n = nW + 8 * cc;
                    if (m < ep.Mtot && n < ep.Ntot) {
                        int64_t offD, offC;
                        ep.offsets(pe, m, n, offD, offC);
                        __builtin_nontemporal_store(v, (HGlbS8)(uintptr_t)(ep.D + offD));
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {             // four passes of 32 rows: the epilogue's image is four 32 x 32 fp32 fragments
#pragma unroll
                for (int F = 0; F < 4; ++F)
#pragma unroll
                    for (int h = 0; h < 4; ++h) {     // 16 x 16 quarter (h >> 1, h & 1) of 32 x 32 fragment F
                        float* st = ep.scratch + F * 1024 + (16 * (h >> 1) + 4 * (laneE >> 4)) * 32 + 16 * (h & 1) + (laneE & 15);
                        const f32x4& c = acc[2 * i + (h >> 1)][2 * F + (h & 1)];
                        st[0] = ep.alpha * c[0]; st[32] = ep.alpha * c[1]; st[64] = ep.alpha * c[2]; st[96] = ep.alpha * c[3];
                    }
                const uint32_t mB = mW + 32 * i;
                ep.template flush<BF, 0>(pe, mB, 0u, 0u, nW, 64u, 32u, laneE);
            }
        }
#endif
        if (!more) break;
        __syncthreads();                          // the epilogue's images in the ring are dead in every wave
        vb = nextVb;
        staged = false;
    }
}

extern "C" int ctamd_h16p_grid_cap;
extern "C" int ctamd_h16p_stagger;

template <bool BF, int LA, int LB>
static hipError_t launch_h16w4p(const GettParams& p, hipStream_t stream) {
    // one workgroup per CU (160 KiB of LDS, 512 registers per lane), each walking tiles blockIdx.x, + grid, + 2 grid, ...
    // the CU count of the CURRENT device (cuTENSORMg / cutensorMp drive several from one process; partitions differ in size): a small
    // per-device table, filled on first use (a benign race: every writer stores the same value)
    static int cuTable[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    int numCUs = (dev >= 0 && dev < 64) ? cuTable[dev] : 0;
    if (numCUs <= 0) {
        if (hipDeviceGetAttribute(&numCUs, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || numCUs <= 0) {
            (void)hipGetLastError();
            numCUs = 256;
        }
        if (dev >= 0 && dev < 64) cuTable[dev] = numCUs;
    }
    if (p.gK.total % (uint32_t)kHBK != 0u) return hipErrorInvalidValue;   // no masked K-tile in this kernel: the planner never hands it a ragged K
    const int cap = ctamd_h16p_grid_cap;          // test hook (CUTENSOR_AMD_H16P_GRID, read by the hooks flavour in api.cpp)
    uint32_t grid = (uint32_t)(cap > 0 ? cap : numCUs);
    grid &= ~7u;                                  // a multiple of the XCD count: tile id % 8 = XCD for every tile of a workgroup
    if (grid == 0) grid = 8;
    if (grid > p.nBlocks) grid = p.nBlocks;
    // Staggered start for short contracted ranges (round 6).  A streamed tile is kt K-tiles of ~2270 cycles that leave the memory side
    // idle and an epilogue of ~12k cycles during which 256 workgroups store 33 MB: started together, the workgroups keep alternating
    // between the two in step.  Four phase groups per XCD, a quarter of a tile's period apart: 8192^2 x 128 / 256 / 512 50.0 / 61.2 / 82.4
    // -> 45.9 / 56.4 / 78.0 us, 16384^2 x 128 160 -> 155; at 16 K-tiles +1.6 % slower, nothing from 32 on: up to 8 K-tiles (profiles/r06zg_*, r06zh_*).  At least four
    // rounds of tiles, or the late starters are the tail.  (CUTENSOR_AMD_H16P_STAGGER, hooks flavour: 0 = off, n = step of n x 1024 cycles)
    GettParams q = p;
    q.mOrg2 = 0u;
    const uint32_t kt = p.kPerSlice / (uint32_t)kHBK;
    if (p.nBlocks >= 4u * grid && p.partial == nullptr) {
        if (ctamd_h16p_stagger > 0) q.mOrg2 = (uint32_t)ctamd_h16p_stagger;
        else if (ctamd_h16p_stagger < 0 && kt <= 8u) q.mOrg2 = (kt * 2270u + 12000u + 2048u) / 4096u;
    }
    hipLaunchKernelGGL((gett_h16w4p_kernel<BF, LA, LB>), dim3(grid), dim3(256), 0, stream, q);
    return hipGetLastError();
}

#define CTAMD_H16W4P_ENTRY(bf, la, lb) \
    {kHTile, kHTile, kHBK, 2, 2, 1, la, lb, 256, 12, 1, 0, &launch_h16w4p<bf, la, lb>, 0},
static const GettKernelInfo g_h16p_table[] = {
    CTAMD_H16W4P_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W4P_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W4P_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W4P_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W4P_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W4P_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W4P_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W4P_ENTRY(false, LAY_F, LAY_F)};

const GettKernelInfo* gett_h16p_kernels(int* count) {
    *count = (int)(sizeof(g_h16p_table) / sizeof(g_h16p_table[0]));
    return g_h16p_table;
}

}  // namespace ctamd
