// gett_h16.hip — bf16 / fp16 GETT kernels for gfx950 (MI355X, CDNA4): 16-bit data, fp32 accumulation
// on v_mfma_f32_32x32x16_{bf16,f16}.  Since round 3 the family's default for long K ranges is gett_h16w4x_kernel in
// gett_h16v.hip (four waves, v_mfma_f32_16x16x32); the eight-wave kernel below serves K ranges of <= 16 K-tiles per workgroup,
// the others are measured alternatives and autotuning candidates.  Shared pieces: gett_h16_common.h.
//
// Reference call sites: cuTENSOR/contraction.cu:261-265 with the types of :33-40 set to 16-bit data
// (BASELINE configs[3]: C[m,n] = sum_k A[m,k] B[k,n], M = N = K = 8192) and the PyTorch binding's
// CUDA_R_16BF / CUTENSOR_COMPUTE_DESC_16BF pairing (python/cutensor/torch/einsum.cc:35-40); alpha and
// beta are fp32 host scalars (einsum.cc:39).
//
// Same GEMM view as gett_f32.hip (mode groups M / N / K / L as mixed-radix numbers, nothing is ever
// transposed in memory); machine mapping:
//
//   * one 512-thread workgroup per CU owns a 256 x 256 output tile; K advances in tiles of 64.
//   * HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, issued from
//     inline asm so that the compiler's wait-count pass does not drain it in front of every ds_read).
//     A K-tile is four 16-KiB half-tiles (A rows 0-127 / 128-255; B half h = the 32-column stripes
//     64 j + 32 h + [0,32), j = 0..3, in the default kernel, columns 128 h + [0,128) in the variants) and the
//     LDS holds two K-tiles = eight half-tile slots (128 KiB).  Which 16-byte unit of the half-tile a lane
//     fetches is free, so the LDS image is shaped by permuting the *source* units:
//       - K-contiguous operand (LAY_K): image [128 rows][64 k] (128-byte rows); unit p of row r holds
//         k-unit p ^ ((r >> 1) & 7): the ds_read_b128 fragment reads (32 rows x 2 k-units) are
//         conflict-free;
//       - free-contiguous operand (LAY_F): image [64 k][128 rows] (256-byte k-rows); unit p of k-row k
//         holds row-unit p ^ (4 * (k & 3)); fragments are read with the transposing ds_read_b64_tr_b16
//         (4 k x 16 rows per 16-lane group), the four k-rows of a half-wave fall into the four 64-byte
//         quarters of the bank row: conflict-free.
//   * 8 waves as 2 (M) x 4 (N); a wave owns rows {64 wr + [0,64)} of both A halves and stripe wc of
//     both B halves (output columns 64 wc + [0,64)): 2 x 2 x 2 accumulator fragments of 32 x 32 (128
//     registers).
//     A K-tile is four phases, one accumulator quadrant (64 x 32, eight MFMAs) each, in the order
//     (a0,b0) (a0,b1) (a1,b1) (a1,b0): a phase reads only the operand half that changes, so A-half 0 and
//     B-half 0 are dead after phase 0, B-half 1 after phase 1, A-half 1 after phase 2 — each slot is
//     restaged (one to two K-tiles ahead) two or three phases after its last read, one half-tile per
//     phase, and four half-tiles (64 KiB per CU) are always in flight behind a *counted* vmcnt.
//   * the two wave rows run half a phase apart (one extra barrier at the start for wr = 1): while the
//     four waves of one row issue their eight MFMAs, the four waves of the other row (one per SIMD
//     each) read fragments and issue LDS-DMA — matrix pipe beside memory pipe on every SIMD.
//
// Roofline: bf16/fp16 MFMA (4096 flop/clk/CU dense); algorithmic flops = 2*L*M*N*K, algorithmic bytes =
// |A| + |B| + |D| (16-bit elements).
#include "gett_h16_common.h"

namespace ctamd {

// The five kernel families of this file are RETIRED defaults (rounds 1-3: the eight-wave ping-pong kernel and its four-wave / streamed /
// register-staged siblings): since round 5 they are compiled only into research builds (make RESEARCH=1 -> -DCTAMD_RESEARCH_KERNELS);
// the production library keeps their table slots (the planner's indices do not move) with a launcher that answers
// hipErrorNotSupported and `ablation` = 2 ("not built"), which rank_h16_choices skips.  The MFMA-only rate measurement below stays.
#if defined(CTAMD_RESEARCH_KERNELS)
// TIMED (measurement-only instantiation, selected with CUTENSOR_AMD_H16_TIMED=1): waves 0 and 4 of workgroup 0
// record s_memtime at the segment boundaries of K-tile 8 into p.timing (7 stamps x 4 phases per wave).
// ABL (measurement only, wrong results; CUTENSOR_AMD_H16_ABL with the default kernel): 1 = no LDS-DMA in the main loop,
// 2 = no A-fragment reads in the main loop, 3 = neither fragment reads nor LDS-DMA (MFMAs + barriers only),
// 4 = no B-fragment reads, 8 = no B traffic through LDS at all (neither LDS-DMA nor fragment reads), 5 = complete main loop
// but no epilogue — what each kind of data movement costs under the power
// limit on random data, and what the epilogue costs.
template <bool BF, int LA, int LB, bool TIMED = false, int ABL = 0>
__global__ void __launch_bounds__(512, 2) gett_h16_kernel(const GettParams p) {
    // operand ring: 8 half-tile slots (slot index: buffer * 4 + {0: A-half 0, 1: A-half 1, 2: B-half 0, 3: B-half 1});
    // the epilogue turns the result through the first 4 KiB per wave of it
    __shared__ __attribute__((aligned(16))) char lds[8 * kHalfBytes];
    unsigned long long wgStamp[6] = {0, 0, 0, 0, 0, 0};   // TIMED: cycles at entry / loop start / loop end / exit, wall clock at entry / exit
    prefetch_kernarg<(int)sizeof(GettParams)>();

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    uint32_t offK[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) offK[s] = h_offK(lane, s);
    const uint32_t offFa0 = h_offF(lane, 2 * wr), offFa1 = h_offF(lane, 2 * wr + 1), offFb = h_offF(lane, wc);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;   // LDS byte address of the ring
    // The argument block is re-read (scalar loads through a laundered kernarg pointer) wherever a tile boundary needs it:
    // kept in SGPRs across the main loop, the mode tables alone would spill a few hundred scalars into VGPR lanes.

    // ---- state of the tile being staged / computed -------------------------------------------------------------------
    uint32_t m0 = 0, n0 = 0, l = 0, slice = 0;
    int nTiles = 0;
    uint64_t bA = 0, bB = 0;
    HOperand<LA> oa;
    HOperand<LB, 8, true> ob;                      // B halves = interleaved 32-column stripes (see HOperand)
    HOdometer odo;
    uint64_t offA1 = 0, offB1 = 0, offA2 = 0, offB2 = 0;
    int tNext = 0;

    // Tile mapping: XCD-contiguous ids, then groups of 8 tile rows so that the 32 tiles an XCD runs at a time form an 8 x 4
    // block (12 operand panels for 32 tiles).
    auto setup = [&](uint32_t vb, const GettParams& p) {
        const uint32_t kTilesAll = p.gK.total / kHBK, tilesPerSlice = p.kPerSlice / kHBK;
        const uint32_t tilesMN = p.tilesM * p.tilesN;
        const uint32_t tilesAll = tilesMN * p.gL.total;
        uint32_t id = xcd_remap(vb, p.nBlocks);
        slice = id / tilesAll;                     // split-K: slice-major ids (splitK == 1: slice = 0)
        id -= slice * tilesAll;
        l = id / tilesMN;
        id -= l * tilesMN;
        const uint32_t perGroup = 8u * p.tilesN;
        const uint32_t grp = id / perGroup, inGrp = id - grp * perGroup;
        const uint32_t first = grp * 8u;
        const uint32_t gsz = (p.tilesM - first < 8u) ? (p.tilesM - first) : 8u;
        const uint32_t mt = first + inGrp % gsz, nt = inGrp / gsz;
        m0 = mt * kHTile; n0 = nt * kHTile;
        // K range of this slice in K-tiles (fast-K: every tile is full)
        const uint32_t tile0 = slice * tilesPerSlice;
        nTiles = (int)((tile0 + tilesPerSlice <= kTilesAll) ? tilesPerSlice : (kTilesAll - tile0));
        bA = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)));
        bB = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)));
        int laneT = lane;                          // opaque: nothing derived from it may be hoisted out of the tile loop and kept
        asm volatile("" : "+v"(laneT));            // (spilled) across the main loop
        oa.init(p.gM, p.gK.stride[0][0], m0, wave, laneT);
        ob.init(p.gN, p.gK.stride[1][0], n0, wave, laneT);
        bA += oa.base;                             // descriptor base = operand + batch offset + this wave's smallest piece offset
        bB += ob.base;
        odo.init(p.gK, tile0 * kHBK);
    };
    // ---- prologue: K-tile 0 and the first halves of K-tile 1, in consumption order ---------------------
    // Staging schedule (global phase P = 4 t + q; one half-tile per phase, restaged two or more phases
    // after the slot's last read, so a slot is never rewritten while the other wave row may still have
    // reads of it in flight):
    //   q = 0: B-half 1 of tile t + 1     q = 1: A-half 1 of tile t + 1
    //   q = 2: A-half 0 of tile t + 2     q = 3: B-half 0 of tile t + 2
    // A half-tile is first read at least five phases after it was issued and the wait in front of each
    // phase's first barrier leaves four half-tiles (8 LDS-DMA instructions of this wave) in flight.
    auto prologue = [&]() {
        oa.issue(h_make_rsrc(bA + odo.offA), 0, ldsBase + 0 * kHalfBytes, wave);
        ob.issue(h_make_rsrc(bB + odo.offB), 0, ldsBase + 2 * kHalfBytes, wave);
        ob.issue(h_make_rsrc(bB + odo.offB), 1, ldsBase + 3 * kHalfBytes, wave);
        oa.issue(h_make_rsrc(bA + odo.offA), 1, ldsBase + 1 * kHalfBytes, wave);
        if (1 < nTiles) odo.advance(p.gK);
        offA1 = odo.offA; offB1 = odo.offB;        // offsets of tile t + 1 (t = current tile)
        oa.issue(h_make_rsrc(bA + offA1), 0, ldsBase + 4 * kHalfBytes, wave);
        ob.issue(h_make_rsrc(bB + offB1), 0, ldsBase + 6 * kHalfBytes, wave);
        tNext = 2;                                 // K-tile the odometer is about to describe
        if (tNext < nTiles) odo.advance(p.gK);
        offA2 = odo.offA; offB2 = odo.offB;        // offsets of tile t + 2
    };

    f32x16 acc[2][2][2];                          // [A half][32-row fragment][B half]
    s16x8 a[2][4], b0[4], b1[4];

    // rows of this wave inside an A half-tile: 64 wr + 32 fa; columns inside a B half-tile: 32 wc
#define CTAMD_H_READ_A(SLOT)                                                                       \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                \
        a[0][s] = h_read_frag<LA>(lds + (SLOT) * kHalfBytes, 64 * wr, s, offK, offFa0);            \
        a[1][s] = h_read_frag<LA>(lds + (SLOT) * kHalfBytes, 64 * wr + 32, s, offK, offFa1);       \
    }
#define CTAMD_H_READ_B(SLOT, DST)                                                                  \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) DST[s] = h_read_frag<LB>(lds + (SLOT) * kHalfBytes, 32 * wc, s, offK, offFb);
    // Register fences (empty asm statements with "+v" operands): hipcc moves register-only MFMAs across
    // barriers, waits and sched_barrier alike; making the operands opaque right after the barrier and the
    // accumulators opaque right after the last MFMA pins the eight MFMAs of a phase inside its segment.
#define CTAMD_H_FENCE_A() asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), \
                                            "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
#define CTAMD_H_FENCE_B(BREG) asm volatile("" : "+v"(BREG[0]), "+v"(BREG[1]), "+v"(BREG[2]), "+v"(BREG[3]));
#define CTAMD_H_FENCE_ACC(AH, BH) asm volatile("" : "+v"(acc[AH][0][BH]), "+v"(acc[AH][1][BH]));
#define CTAMD_H_MFMA_RANGE(AH, BH, BREG, S0, S1)                                                   \
    _Pragma("unroll") for (int s = (S0); s < (S1); ++s) {                                          \
        acc[AH][0][BH] = h_mfma<BF>(a[0][s], BREG[s], acc[AH][0][BH]);                             \
        acc[AH][1][BH] = h_mfma<BF>(a[1][s], BREG[s], acc[AH][1][BH]);                             \
    }
#define CTAMD_H_MFMA_Q(Q, S0, S1)                                                                  \
        if constexpr ((Q) == 0) { CTAMD_H_MFMA_RANGE(0, 0, b0, S0, S1) CTAMD_H_FENCE_ACC(0, 0) }   \
        if constexpr ((Q) == 1) { CTAMD_H_MFMA_RANGE(0, 1, b1, S0, S1) CTAMD_H_FENCE_ACC(0, 1) }   \
        if constexpr ((Q) == 2) { CTAMD_H_MFMA_RANGE(1, 1, b1, S0, S1) CTAMD_H_FENCE_ACC(1, 1) }   \
        if constexpr ((Q) == 3) { CTAMD_H_MFMA_RANGE(1, 0, b0, S0, S1) CTAMD_H_FENCE_ACC(1, 0) }

    // One phase.  P = LDS buffer of the current K-tile (compile time), Q = phase inside the tile.
    //   load segment : fragment reads of the operand half that changes, one half-tile of LDS-DMA, counted
    //                  wait for the half-tile the NEXT phase reads (four stay in flight)
    //   barrier      : the other wave row has finished its MFMA segment / issued its half of the DMA
    //   MFMA segment : own fragment reads are back (their latency overlapped the barrier), 8 x 32x32x16 on
    //                  one accumulator quadrant
    //   barrier
#define CTAMD_H_STAMP(Q, I)                                                                        \
    if constexpr (TIMED) {                                                                         \
        if (tstamp && t8 == 8 && lane == 0) tstamp[(Q) * 7 + (I)] = __builtin_readcyclecounter();  \
    }
#define CTAMD_H_PHASE(P, Q)                                                                        \
    {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        CTAMD_H_STAMP(Q, 0)                                                                        \
        if constexpr ((Q) == 0 && ABL != 2 && ABL != 3) { CTAMD_H_READ_A((P) * 4 + 0) }             \
        if constexpr ((Q) == 0 && ABL != 4 && ABL != 3 && ABL != 8) { CTAMD_H_READ_B((P) * 4 + 2, b0) }         \
        if constexpr ((Q) == 1 && ABL != 4 && ABL != 3 && ABL != 8) { CTAMD_H_READ_B((P) * 4 + 3, b1) }         \
        if constexpr ((Q) == 2 && ABL != 2 && ABL != 3) { CTAMD_H_READ_A((P) * 4 + 1) }             \
        if constexpr ((Q) == 0 && ABL != 1 && ABL != 3 && ABL != 8) ob.template issue<false>(h_make_rsrc(bB + offB1), 1, ldsBase + (((P) ^ 1) * 4 + 3) * kHalfBytes, wave);  \
        if constexpr ((Q) == 1 && ABL != 1 && ABL != 3) oa.template issue<false>(h_make_rsrc(bA + offA1), 1, ldsBase + (((P) ^ 1) * 4 + 1) * kHalfBytes, wave);  \
        if constexpr ((Q) == 2 && ABL != 1 && ABL != 3) oa.template issue<false>(h_make_rsrc(bA + offA2), 0, ldsBase + ((P) * 4 + 0) * kHalfBytes, wave);        \
        if constexpr ((Q) == 3) {                                                                  \
            if constexpr (ABL != 1 && ABL != 3 && ABL != 8) ob.template issue<false>(h_make_rsrc(bB + offB2), 0, ldsBase + ((P) * 4 + 2) * kHalfBytes, wave);    \
            offA1 = offA2; offB1 = offB2;                                                          \
            ++tNext;                                                                               \
            if (tNext < nTiles) odo.advance(p.gK);   /* past the end: re-stage the last tile (never read) */ \
            offA2 = odo.offA; offB2 = odo.offB;                                                    \
        }                                                                                          \
        CTAMD_H_STAMP(Q, 1)                                                                        \
        CTAMD_H_VMCNT(8);                                                                          \
        CTAMD_H_STAMP(Q, 2)                                                                        \
        __builtin_amdgcn_s_barrier();                                                              \
        CTAMD_H_STAMP(Q, 3)                                                                        \
        CTAMD_H_LGKM0();                                                                           \
        CTAMD_H_STAMP(Q, 4)                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        CTAMD_H_FENCE_A()                                                                          \
        if constexpr ((Q) == 0 || (Q) == 3) { CTAMD_H_FENCE_B(b0) } else { CTAMD_H_FENCE_B(b1) }   \
        __builtin_amdgcn_s_setprio(1);                                                             \
        CTAMD_H_MFMA_Q(Q, 0, 4)                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        CTAMD_H_STAMP(Q, 5)                                                                        \
        __builtin_amdgcn_s_barrier();                                                              \
        CTAMD_H_STAMP(Q, 6)                                                                        \
    }
#define CTAMD_H_TILE(P) CTAMD_H_PHASE(P, 0) CTAMD_H_PHASE(P, 1) CTAMD_H_PHASE(P, 2) CTAMD_H_PHASE(P, 3)

    unsigned long long* tstamp = nullptr;
    int t8 = 0;
    if constexpr (TIMED) {
        if (p.timing != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 4)) tstamp = p.timing + (wave >> 2) * 32;
    }

    // One output tile per workgroup.  (A persistent form — one workgroup per CU walking over its tiles, the next tile's first
    // half-tiles streaming in while the result goes out — measured SLOWER, 529 vs 505 us on 8192^3 zeros: vmcnt counts loads and
    // stores in order, so the next tile's main loop waits for the stores anyway, whereas a workgroup that ends leaves its stores
    // to drain behind the next workgroup's start.)
    setup(blockIdx.x, p);
    prologue();
    {
        if constexpr (TIMED) { wgStamp[0] = __builtin_readcyclecounter(); wgStamp[4] = wall_clock64(); }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int z = 0; z < 2; ++z)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[x][y][z][r] = 0.f;
        CTAMD_H_VMCNT(8);                             // A-half 0 and B-half 0 of tile 0 have landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();    // second wave row runs half a phase behind

        if constexpr ((ABL >= 2 && ABL <= 5) || ABL == 8) {   // the fragments the ablated loop never refreshes
            CTAMD_H_READ_A(0) CTAMD_H_READ_B(2, b0) CTAMD_H_READ_B(3, b1)
        }
        if constexpr (TIMED) wgStamp[1] = __builtin_readcyclecounter();
        int t = 0;
        for (; t + 1 < nTiles; t += 2) { t8 = t; CTAMD_H_TILE(0) t8 = t + 1; CTAMD_H_TILE(1) }
        if (t < nTiles) { t8 = t; CTAMD_H_TILE(0) }
        if constexpr (TIMED) wgStamp[2] = __builtin_readcyclecounter();
        if (wr == 0) __builtin_amdgcn_s_barrier();    // pairs with the last barrier of the second wave row
        CTAMD_H_VMCNT(0);                             // the re-staged tail tiles: no LDS-DMA may outlive its tile
        __syncthreads();                              // every wave has finished reading the ring, every piece has landed

        const uint32_t em0 = m0, en0 = n0, el = l, eslice = slice, evb = blockIdx.x;
        GettParams q;                                 // a fresh copy of the arguments for the epilogue (only the fields used are loaded)
        h_reload_params(q);

        if (q.partial != nullptr) {
            // ---- split-K: fp32 partial tile, row-major [slice][l][m][n] (32 lanes x 4 B contiguous along n); the fold
            //      (splitk_reduce_kernel, 16-bit output) applies alpha / beta ---------------------------------------
            int laneE = lane;
            asm volatile("" : "+v"(laneE));
            const uint32_t Mt = q.gM.total, Nt = q.gN.total;
            float* P = q.partial + ((size_t)eslice * q.gL.total + el) * (size_t)Mt * Nt;
            auto store_partial = [&](const f32x16& c0, const f32x16& c1, uint32_t mBase) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t m = mBase + (r & 3) + 8 * (r >> 2) + 4 * (laneE >> 5);
                    if (m < Mt) {
                        const uint32_t na = en0 + 64 * wc + (laneE & 31), nb = na + 32;
                        if (na < Nt) P[(size_t)m * Nt + na] = c0[r];
                        if (nb < Nt) P[(size_t)m * Nt + nb] = c1[r];
                    }
                }
            };
            store_partial(acc[0][0][0], acc[0][0][1], em0 + 64 * wr);
            store_partial(acc[0][1][0], acc[0][1][1], em0 + 64 * wr + 32);
            store_partial(acc[1][0][0], acc[1][0][1], em0 + 128 + 64 * wr);
            store_partial(acc[1][1][0], acc[1][1][1], em0 + 128 + 64 * wr + 32);
        } else if constexpr (ABL == 5) {              // measurement only: no epilogue (one word per lane keeps the accumulators alive)
            float keep = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) keep += acc[0][0][0][r] + acc[0][0][1][r] + acc[0][1][0][r] + acc[0][1][1][r] + acc[1][0][0][r] + acc[1][0][1][r] + acc[1][1][0][r] + acc[1][1][1][r];
            if (keep == 12345.678f) static_cast<uint16_t*>(q.D)[tid] = 1;
        } else {
            // ---- epilogue: D = alpha * acc + beta * C, 16 rows x 64 columns at a time through 4 KiB of LDS per wave --------
            HEpilogue ep;
            ep.init(q, el, lds, wave, 4096);
            int laneE = lane;                         // opaque (see setup): lane-derived addresses are recomputed here, not kept
            asm volatile("" : "+v"(laneE));
            constexpr int ST = (ABL == 6 ? 1 : ABL == 7 ? 2 : 0);
            const uint32_t nB = en0 + 64 * wc;
            // ONE copy of the store code: a rolled loop over the eight (fragment pair, half) passes; only the parking of the
            // accumulators (registers cannot be indexed at run time) is selected by a wave-uniform switch
#pragma unroll 1
            for (int pass = 0; pass < 8; ++pass) {
                switch (pass) {
                    case 0: ep.template park_pair<0>(acc[0][0][0], acc[0][0][1], laneE); break;
                    case 1: ep.template park_pair<1>(acc[0][0][0], acc[0][0][1], laneE); break;
                    case 2: ep.template park_pair<0>(acc[0][1][0], acc[0][1][1], laneE); break;
                    case 3: ep.template park_pair<1>(acc[0][1][0], acc[0][1][1], laneE); break;
                    case 4: ep.template park_pair<0>(acc[1][0][0], acc[1][0][1], laneE); break;
                    case 5: ep.template park_pair<1>(acc[1][0][0], acc[1][0][1], laneE); break;
                    case 6: ep.template park_pair<0>(acc[1][1][0], acc[1][1][1], laneE); break;
                    default: ep.template park_pair<1>(acc[1][1][0], acc[1][1][1], laneE); break;
                }
                const uint32_t mB = em0 + 128u * (uint32_t)(pass >> 2) + 64u * (uint32_t)wr + 16u * (uint32_t)(pass & 3);
                ep.template flush_pair<BF, ST>(q, mB, nB, laneE);
            }
        }
        if constexpr (TIMED) {
            if (p.timing != nullptr && tid == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores of this wave have left
                wgStamp[3] = __builtin_readcyclecounter();
                wgStamp[5] = wall_clock64();
#pragma unroll
                for (int i = 0; i < 6; ++i) p.timing[64 + 8 * (size_t)evb + i] = wgStamp[i];
                p.timing[64 + 8 * (size_t)evb + 6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;   // HW_REG_XCC_ID
            }
        }
    }
}


// =====================================================================================================
// Four-wave variant (one wave per SIMD, 512 registers per lane): same 256 x 256 x 64 tile, LDS images and
// source-side swizzles as above; 4 waves as 2 (M) x 2 (N), a wave owns a 128 x 128 quadrant = 4 x 4
// accumulator fragments (256 registers).  Per K-tile a wave issues 64 MFMAs, reads 32 fragments (half the LDS
// traffic per MFMA of the 2 x 4 arrangement) and stages 16 KiB by LDS-DMA; nothing is handed from wave to
// wave, so there is ONE barrier per K-tile instead of eight:
//   k-steps 0..2 : fragment reads of step s + 1 interleaved with the 16 MFMAs of step s (two register sets);
//   k-step  3    : own reads of this buffer are back, own pieces of tile t + 1 have landed -> barrier; now
//                  the buffer of tile t is free and the buffer of tile t + 1 complete: first fragment reads of
//                  tile t + 1 and the 16 LDS-DMA pieces of tile t + 2 (one per MFMA) beside the last 16 MFMAs.
// The matrix pipe only waits for the skew between the four waves at that barrier.
// =====================================================================================================
// ABL (measurement only, wrong results): 1 = no LDS-DMA in the main loop, 2 = LDS-DMA issued but never waited for,
// 3 = no barrier / waits at the tile boundary.
template <bool BF, int LA, int LB, int ABL = 0>
__global__ void __launch_bounds__(256, 1) gett_h16w4_kernel(const GettParams p) {
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

    uint64_t bA = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)));
    uint64_t bB = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)));
    HOperand<LA, 4> oa;
    HOperand<LB, 4> ob;
    oa.init(p.gM, p.gK.stride[0][0], m0, wave, lane);
    ob.init(p.gN, p.gK.stride[1][0], n0, wave, lane);
    bA += oa.base;                                 // descriptor base = operand + batch offset + this wave's smallest piece offset
    bB += ob.base;

    uint32_t offK[4], offF[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { offK[s] = h_offK(lane, s); offF[s] = h_offF(lane, s); }

    HOdometer odo;
    odo.init(p.gK, tile0 * kHBK);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // piece n = 0..15 of a K-tile for this wave: operand half q = n >> 2 (A0, A1, B0, B1), piece i = n & 3
#define CTAMD_W4_DMA(P, N, PAD)                                                                                     \
    {                                                                                                              \
        constexpr int q_ = (N) >> 2, i_ = (N) & 3;                                                                 \
        if constexpr (q_ < 2) oa.template issue_piece<PAD>(h_make_rsrc(bA + odo.offA), q_, i_, ldsBase + ((P) * 4 + q_) * kHalfBytes, wave); \
        else ob.template issue_piece<PAD>(h_make_rsrc(bB + odo.offB), q_ - 2, i_, ldsBase + ((P) * 4 + q_) * kHalfBytes, wave);   \
    }
#define CTAMD_W4_STAGE(P, PAD)                                                                                      \
    CTAMD_W4_DMA(P, 0, PAD) CTAMD_W4_DMA(P, 1, PAD) CTAMD_W4_DMA(P, 2, PAD) CTAMD_W4_DMA(P, 3, PAD)                \
    CTAMD_W4_DMA(P, 4, PAD) CTAMD_W4_DMA(P, 5, PAD) CTAMD_W4_DMA(P, 6, PAD) CTAMD_W4_DMA(P, 7, PAD)                \
    CTAMD_W4_DMA(P, 8, PAD) CTAMD_W4_DMA(P, 9, PAD) CTAMD_W4_DMA(P, 10, PAD) CTAMD_W4_DMA(P, 11, PAD)              \
    CTAMD_W4_DMA(P, 12, PAD) CTAMD_W4_DMA(P, 13, PAD) CTAMD_W4_DMA(P, 14, PAD) CTAMD_W4_DMA(P, 15, PAD)

    // ---- prologue: K-tiles 0 and 1 ------------------------------------------------------------------------
    CTAMD_W4_STAGE(0, true)
    if (1 < nTiles) odo.advance(p.gK);           // past the end the last tile is re-staged (never read)
    CTAMD_W4_DMA(1, 0, true) CTAMD_W4_DMA(1, 1, true) CTAMD_W4_DMA(1, 2, true) CTAMD_W4_DMA(1, 3, true)
    CTAMD_W4_DMA(1, 4, true) CTAMD_W4_DMA(1, 5, true) CTAMD_W4_DMA(1, 6, true) CTAMD_W4_DMA(1, 7, true)
    int tNext = 1;                                // K-tile the odometer describes (its second half goes out in k-step 0)
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

    const char* const aSlot0 = lds + wr * kHalfBytes;          // A-half wr of buffer 0 (buffer 1: + 4 slots)
    const char* const bSlot0 = lds + (2 + wc) * kHalfBytes;    // B-half wc of buffer 0
    // fragment f = 0..7 of k-step S from buffer P into register set SET: f < 4 -> B columns 32 f (all four feed the first
    // MFMAs of a step), else A rows 32 (f - 4)
#define CTAMD_W4_READ(P, S, SET, F)                                                                                 \
    {                                                                                                              \
        if constexpr (ABL >= 4 && (P) < 2) {}                                                                      \
        else if constexpr ((F) < 4) b[SET][F] = h_read_frag<LB>(bSlot0 + (P) * 4 * kHalfBytes, 32 * (F), S, offK, offF[F]);          \
        else a[SET][(F) - 4] = h_read_frag<LA>(aSlot0 + (P) * 4 * kHalfBytes, 32 * ((F) - 4), S, offK, offF[(F) - 4]);        \
    }
#define CTAMD_W4_MFMA(SET, M) acc[(M) >> 2][(M) & 3] = h_mfma<BF>(a[SET][(M) >> 2], b[SET][(M) & 3], acc[(M) >> 2][(M) & 3]);
    // k-step S < 3: one fragment read of step S + 1 per two MFMAs of step S; k-step 0 also carries the second half
    // (pieces 8..15) of the tile being staged into the other buffer
#define CTAMD_W4_PAIR(P, S, F)                                                                                      \
    CTAMD_W4_READ(P, (S) + 1, ((S) + 1) & 1, F) CTAMD_W4_MFMA((S) & 1, 2 * (F))                                    \
    if constexpr ((S) == 0 && ABL != 1 && ABL != 4) CTAMD_W4_DMA((P) ^ 1, 8 + (F), false)                          \
    CTAMD_W4_MFMA((S) & 1, 2 * (F) + 1)                                                                            \
    __builtin_amdgcn_sched_barrier(0);
#define CTAMD_W4_STEP(P, S)                                                                                         \
    CTAMD_W4_PAIR(P, S, 0) CTAMD_W4_PAIR(P, S, 1) CTAMD_W4_PAIR(P, S, 2) CTAMD_W4_PAIR(P, S, 3)                    \
    CTAMD_W4_PAIR(P, S, 4) CTAMD_W4_PAIR(P, S, 5) CTAMD_W4_PAIR(P, S, 6) CTAMD_W4_PAIR(P, S, 7)
    // k-step 3: reads of the next tile's step 0 (other buffer), the first half (pieces 0..7) of tile t + 2 into this
    // buffer, MFMAs of step 3 — a read and a piece alternate, one per MFMA
#define CTAMD_W4_LAST2(P, F)                                                                                        \
    CTAMD_W4_READ((P) ^ 1, 0, 0, F)                                                                                \
    CTAMD_W4_MFMA(1, 2 * (F)) __builtin_amdgcn_sched_barrier(0);                                                   \
    if constexpr (ABL != 1 && ABL != 4) CTAMD_W4_DMA(P, F, false)                                                  \
    CTAMD_W4_MFMA(1, 2 * (F) + 1) __builtin_amdgcn_sched_barrier(0);
#define CTAMD_W4_TILE(P)                                                                                            \
    CTAMD_W4_STEP(P, 0)                                                                                            \
    ++tNext;                                                                                                       \
    if (tNext < nTiles) odo.advance(p.gK);                                                                         \
    CTAMD_W4_STEP(P, 1) CTAMD_W4_STEP(P, 2)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    if constexpr (ABL != 3) CTAMD_H_LGKM0();                                                                       \
    if constexpr (ABL != 2 && ABL != 3) CTAMD_H_VMCNT(0);                                                          \
    if constexpr (ABL != 3) __builtin_amdgcn_s_barrier();                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_W4_LAST2(P, 0) CTAMD_W4_LAST2(P, 1) CTAMD_W4_LAST2(P, 2) CTAMD_W4_LAST2(P, 3)                            \
    CTAMD_W4_LAST2(P, 4) CTAMD_W4_LAST2(P, 5) CTAMD_W4_LAST2(P, 6) CTAMD_W4_LAST2(P, 7)

    // first fragments of tile 0
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        b[0][f] = h_read_frag<LB>(bSlot0, 32 * f, 0, offK, offF[f]);
        a[0][f] = h_read_frag<LA>(aSlot0, 32 * f, 0, offK, offF[f]);
        if constexpr (ABL >= 4) { b[1][f] = b[0][f]; a[1][f] = a[0][f]; }
    }
    int t = 0;
    for (; t + 1 < nTiles; t += 2) { CTAMD_W4_TILE(0) CTAMD_W4_TILE(1) }
    if (t < nTiles) { CTAMD_W4_TILE(0) }
    CTAMD_H_VMCNT(0);                             // the re-staged tail: no LDS-DMA may outlive the workgroup

    const uint32_t mW = m0 + 128 * wr, nW = n0 + 128 * wc;    // this wave's quadrant
    if (p.partial != nullptr) {                   // split-K: fp32 partial tile, row-major [slice][l][m][n]
        const uint32_t Mt = p.gM.total, Nt = p.gN.total;
        float* P = p.partial + ((size_t)slice * p.gL.total + l) * (size_t)Mt * Nt;
        auto store_partial = [&](const f32x16& c0, const f32x16& c1, const f32x16& c2, const f32x16& c3, uint32_t mBase) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = mBase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < Mt) {
                    const uint32_t n = nW + (lane & 31);
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
        ep.park(0, acc[i][0], lane); ep.park(1, acc[i][1], lane); ep.park(2, acc[i][2], lane); ep.park(3, acc[i][3], lane);
        const uint32_t mB = mW + 32 * i;
        ep.template flush<BF>(p, mB, 0u, 0u, nW, 64u, 32u, lane);
    }
}


// =====================================================================================================
// Four-wave REGISTER-STAGED variant (CUTENSOR_AMD_H16_WAVES=4r; round 3): the tile, LDS images, fragment reads and epilogue of
// gett_h16w4_kernel, but the operands reach LDS through registers — buffer_load_dwordx4 into a staging set, ds_write_b128 a
// K-tile later — instead of by LDS-DMA.  Why: with ONE wave per SIMD every instruction the wave issues sits in front of its own
// MFMAs, and an LDS-DMA instruction holds the issue port for 60-100+ cycles (gett_f32_stream.hip header) against the 32 cycles an
// MFMA covers: gett_h16w4_kernel loses 21 % of its cycles to the 16 pieces per K-tile even when nothing waits for them
// (ablation 2, profiles/r03_h16_w4_status.txt), and a deeper ring does not help (gett_h16w4s_kernel, profiles/r03_h16_w4s_status.txt).
// A plain buffer load and a ds_write issue in a few cycles each.  A lane loads exactly the 16-byte unit the LDS-DMA form would
// (same HOperand source offsets) and writes it where the DMA would have put it (piece base + 16 * lane), so the LDS images are
// identical.  Two staging sets of 16 pieces (128 registers of the wave's 512): while the set of tile t + 1 is written to the
// free buffer (k-steps 1, 2 of tile t), the loads of tile t + 2 are already in flight into the other set (k-steps 0, 1) —
// five k-steps (1.1-1.7 us) ahead of their ds_write.  All waits are the compiler's (no LDS-DMA, nothing hidden from its
// counters); ONE barrier per K-tile.
// =====================================================================================================
template <bool BF, int LA, int LB>
__global__ void __launch_bounds__(256, 1) gett_h16w4r_kernel(const GettParams p) {
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

    uint64_t bA = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)));
    uint64_t bB = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)));
    HOperand<LA, 4> oa;
    HOperand<LB, 4> ob;
    oa.init(p.gM, p.gK.stride[0][0], m0, wave, lane);
    ob.init(p.gN, p.gK.stride[1][0], n0, wave, lane);
    bA += oa.base;
    bB += ob.base;

    uint32_t offK[4], offF[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { offK[s] = h_offK(lane, s); offF[s] = h_offF(lane, s); }

    HOdometer odo;
    odo.init(p.gK, tile0 * kHBK);

    typedef uint32_t stg_t __attribute__((ext_vector_type(4)));
    stg_t stg[2][16];                              // two staging sets: piece n = 0..15 of a K-tile (half q = n >> 2, piece i = n & 3)
#if defined(__HIP_DEVICE_COMPILE__)
#define CTAMD_R_LOAD(SET, N)                                                                                        \
    {                                                                                                              \
        constexpr int q_ = (N) >> 2, i_ = (N) & 3;                                                                 \
        if constexpr (q_ < 2)                                                                                      \
            stg[SET][N] = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(bA + odo.offA), 0, -1, 0x00020000), \
                                                                (int)oa.src[q_][i_], 0, 0);                        \
        else                                                                                                       \
            stg[SET][N] = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(bB + odo.offB), 0, -1, 0x00020000), \
                                                                (int)ob.src[q_ - 2][i_], 0, 0);                    \
    }
#else
#define CTAMD_R_LOAD(SET, N) { stg[SET][N] = stg_t{0u, 0u, 0u, 0u}; }
#endif
    // piece N of the K-tile in staging set SET -> ring buffer P, where the LDS-DMA form would have put it
#define CTAMD_R_STORE(P, SET, N)                                                                                    \
    {                                                                                                              \
        constexpr int q_ = (N) >> 2, i_ = (N) & 3;                                                                 \
        *reinterpret_cast<stg_t*>(lds + ((P) * 4 + q_) * kHalfBytes + (wave + 4 * i_) * 1024 + lane * 16) = stg[SET][N]; \
    }
#define CTAMD_R_LOAD16(SET)                                                                                          \
    CTAMD_R_LOAD(SET, 0) CTAMD_R_LOAD(SET, 1) CTAMD_R_LOAD(SET, 2) CTAMD_R_LOAD(SET, 3) CTAMD_R_LOAD(SET, 4) CTAMD_R_LOAD(SET, 5)        \
    CTAMD_R_LOAD(SET, 6) CTAMD_R_LOAD(SET, 7) CTAMD_R_LOAD(SET, 8) CTAMD_R_LOAD(SET, 9) CTAMD_R_LOAD(SET, 10) CTAMD_R_LOAD(SET, 11)     \
    CTAMD_R_LOAD(SET, 12) CTAMD_R_LOAD(SET, 13) CTAMD_R_LOAD(SET, 14) CTAMD_R_LOAD(SET, 15)
#define CTAMD_R_STORE16(P, SET)                                                                                      \
    CTAMD_R_STORE(P, SET, 0) CTAMD_R_STORE(P, SET, 1) CTAMD_R_STORE(P, SET, 2) CTAMD_R_STORE(P, SET, 3) CTAMD_R_STORE(P, SET, 4)        \
    CTAMD_R_STORE(P, SET, 5) CTAMD_R_STORE(P, SET, 6) CTAMD_R_STORE(P, SET, 7) CTAMD_R_STORE(P, SET, 8) CTAMD_R_STORE(P, SET, 9)        \
    CTAMD_R_STORE(P, SET, 10) CTAMD_R_STORE(P, SET, 11) CTAMD_R_STORE(P, SET, 12) CTAMD_R_STORE(P, SET, 13) CTAMD_R_STORE(P, SET, 14)   \
    CTAMD_R_STORE(P, SET, 15)

    // ---- prologue: tile 0 -> set 0 -> buffer 0; tile 1 -> set 1 (stays in registers until tile 0's k-steps 1, 2) ----
    int tNext = 0;                                 // K-tile the odometer describes
    CTAMD_R_LOAD16(0)
    ++tNext;
    if (tNext < nTiles) odo.advance(p.gK);         // past the end the last tile is loaded again (never multiplied)
    CTAMD_R_LOAD16(1)
    ++tNext;
    if (tNext < nTiles) odo.advance(p.gK);
    CTAMD_R_STORE16(0, 0)
    CTAMD_H_LGKM0();
    __builtin_amdgcn_s_barrier();

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    s16x8 a[2][4], b[2][4];                       // two register sets: k-step s uses set s & 1

    const char* const aSlot0 = lds + wr * kHalfBytes;          // A-half wr of buffer 0 (buffer 1: + 4 slots)
    const char* const bSlot0 = lds + (2 + wc) * kHalfBytes;    // B-half wc of buffer 0
#define CTAMD_R_READ(P, S, SET, F)                                                                                  \
    {                                                                                                              \
        if constexpr ((F) < 4) b[SET][F] = h_read_frag<LB>(bSlot0 + (P) * 4 * kHalfBytes, 32 * (F), S, offK, offF[F]);          \
        else a[SET][(F) - 4] = h_read_frag<LA>(aSlot0 + (P) * 4 * kHalfBytes, 32 * ((F) - 4), S, offK, offF[(F) - 4]);        \
    }
#define CTAMD_R_MFMA(SET, M) acc[(M) >> 2][(M) & 3] = h_mfma<BF>(a[SET][(M) >> 2], b[SET][(M) & 3], acc[(M) >> 2][(M) & 3]);
    // one group = two MFMAs of k-step S with one fragment read of the following step and, by k-step, one load / one store / both
    //   KIND 0: load piece F of tile t + 2 (set LS)                 (k-step 0: pieces 0..7)
    //   KIND 1: load piece 8 + F (set LS) and store piece F (set SS -> buffer P ^ 1)   (k-step 1)
    //   KIND 2: store piece 8 + F                                  (k-step 2)
    //   KIND 3: nothing (k-step 3; the read comes from the other buffer)
#define CTAMD_R_GROUP(P, S, F, KIND, LS, SS)                                                                        \
    if constexpr ((S) < 3) { CTAMD_R_READ(P, (S) + 1, ((S) + 1) & 1, F) } else { CTAMD_R_READ((P) ^ 1, 0, 0, F) }  \
    CTAMD_R_MFMA((S) & 1, 2 * (F))                                                                                 \
    if constexpr ((KIND) == 0) { CTAMD_R_LOAD(LS, F) }                                                             \
    if constexpr ((KIND) == 1) { CTAMD_R_LOAD(LS, 8 + (F)) CTAMD_R_STORE((P) ^ 1, SS, F) }                         \
    if constexpr ((KIND) == 2) { CTAMD_R_STORE((P) ^ 1, SS, 8 + (F)) }                                             \
    CTAMD_R_MFMA((S) & 1, 2 * (F) + 1)                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#define CTAMD_R_STEP(P, S, KIND, LS, SS)                                                                            \
    CTAMD_R_GROUP(P, S, 0, KIND, LS, SS) CTAMD_R_GROUP(P, S, 1, KIND, LS, SS) CTAMD_R_GROUP(P, S, 2, KIND, LS, SS) CTAMD_R_GROUP(P, S, 3, KIND, LS, SS) \
    CTAMD_R_GROUP(P, S, 4, KIND, LS, SS) CTAMD_R_GROUP(P, S, 5, KIND, LS, SS) CTAMD_R_GROUP(P, S, 6, KIND, LS, SS) CTAMD_R_GROUP(P, S, 7, KIND, LS, SS)
    // tile in buffer P; staging set P holds nothing live (its tile is in LDS), set P ^ 1 holds tile t + 1
#define CTAMD_R_TILE(P)                                                                                             \
    CTAMD_R_STEP(P, 0, 0, P, (P) ^ 1)                                                                              \
    CTAMD_R_STEP(P, 1, 1, P, (P) ^ 1)                                                                              \
    ++tNext;                                                                                                       \
    if (tNext < nTiles) odo.advance(p.gK);                                                                         \
    CTAMD_R_STEP(P, 2, 2, P, (P) ^ 1)                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_H_LGKM0();                                                                                               \
    __builtin_amdgcn_s_barrier();                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    CTAMD_R_STEP(P, 3, 3, P, (P) ^ 1)

    // first fragments of tile 0
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        b[0][f] = h_read_frag<LB>(bSlot0, 32 * f, 0, offK, offF[f]);
        a[0][f] = h_read_frag<LA>(aSlot0, 32 * f, 0, offK, offF[f]);
    }
    int t = 0;
    for (; t + 1 < nTiles; t += 2) { CTAMD_R_TILE(0) CTAMD_R_TILE(1) }
    if (t < nTiles) { CTAMD_R_TILE(0) }

    const uint32_t mW = m0 + 128 * wr, nW = n0 + 128 * wc;    // this wave's quadrant
    if (p.partial != nullptr) {                   // split-K: fp32 partial tile, row-major [slice][l][m][n]
        const uint32_t Mt = p.gM.total, Nt = p.gN.total;
        float* P = p.partial + ((size_t)slice * p.gL.total + l) * (size_t)Mt * Nt;
        auto store_partial = [&](const f32x16& c0, const f32x16& c1, const f32x16& c2, const f32x16& c3, uint32_t mBase) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = mBase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < Mt) {
                    const uint32_t n = nW + (lane & 31);
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
        ep.park(0, acc[i][0], lane); ep.park(1, acc[i][1], lane); ep.park(2, acc[i][2], lane); ep.park(3, acc[i][3], lane);
        const uint32_t mB = mW + 32 * i;
        ep.template flush<BF>(p, mB, 0u, 0u, nW, 64u, 32u, lane);
    }
}

template <bool BF, int LA, int LB>
static hipError_t launch_h16w4r(const GettParams& p, hipStream_t stream) {
    hipLaunchKernelGGL((gett_h16w4r_kernel<BF, LA, LB>), dim3(p.nBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// Epilogue of the 2 (M) x 4 (N) wave grid with 128 x 64 per wave (gett_h16s_kernel): split-K partial
// tile or D = alpha * acc + beta * C with one rounding to the 16-bit type.
template <bool BF>
__device__ __forceinline__ void h_epilogue_128x64(const GettParams& p, const f32x16 (&acc)[4][2], uint32_t m0, uint32_t n0, int wr, int wc,
                                                  uint32_t slice, uint32_t l, int lane, char* lds, int wave) {
    const uint32_t mW = m0 + 128 * wr, nW = n0 + 64 * wc;     // this wave's 128 x 64 block
    if (p.partial != nullptr) {                   // split-K: fp32 partial tile, row-major [slice][l][m][n]
        const uint32_t Mt = p.gM.total, Nt = p.gN.total;
        float* P = p.partial + ((size_t)slice * p.gL.total + l) * (size_t)Mt * Nt;
        auto store_partial = [&](const f32x16& c0, const f32x16& c1, uint32_t mBase) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = mBase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < Mt) {
                    const uint32_t n = nW + (lane & 31);
                    float* row = P + (size_t)m * Nt;
                    if (n < Nt) row[n] = c0[r];
                    if (n + 32 < Nt) row[n + 32] = c1[r];
                }
            }
        };
        store_partial(acc[0][0], acc[0][1], mW);
        store_partial(acc[1][0], acc[1][1], mW + 32);
        store_partial(acc[2][0], acc[2][1], mW + 64);
        store_partial(acc[3][0], acc[3][1], mW + 96);
        return;
    }
    __syncthreads();                              // every wave has finished reading the operand ring
    HEpilogue ep;
    ep.init(p, l, lds, wave);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                 // two passes: accumulator rows 2 h and 2 h + 1, both column fragments
        ep.park(0, acc[2 * h][0], lane); ep.park(1, acc[2 * h][1], lane); ep.park(2, acc[2 * h + 1][0], lane); ep.park(3, acc[2 * h + 1][1], lane);
        const uint32_t mB = mW + 64 * h;
        ep.template flush<BF>(p, mB, 32u, 0u, nW, 0u, 32u, lane);
    }
}

// =====================================================================================================
// Streamed eight-wave variant (CUTENSOR_AMD_H16_WAVES=s; a measured alternative and an autotuning candidate, not the
// default): the tile and source-side swizzles of the kernels above, 8 waves as 2 (M) x 4 (N), a wave owns 128 x 64 = 4 x 2
// accumulator fragments.  Unlike gett_h16_kernel the two waves of a SIMD are NOT alternated by barriers: every wave
// runs a software pipeline of its own (fragment reads of k-step s + 1 and its LDS-DMA pieces interleaved with the 8
// MFMAs of k-step s, two register sets) and the SIMD's arbiter fills one wave's read / DMA-issue slots with the other
// wave's MFMAs.  K-tile of 32 and an NS-deep LDS ring (NS x 32 KiB): NS - 2 ... NS - 1 K-tiles (64-96 KiB per CU for
// NS = 4, 5) are in flight behind a counted vmcnt.  Per K-tile a wave issues 16 MFMAs (two k-steps), 12 fragment
// reads, 4 LDS-DMA pieces and meets the workgroup ONCE — in front of k-step 1, whose 8 MFMAs are already in registers
// and cover the first fragment reads of the next tile; the tile's buffer is refilled (tile t + NS) behind that barrier.
// Measured: within 0-7 % of gett_h16_kernel on U(-1,1) data depending on the operand layout, 21 %
// behind it on zero-filled operands — the fine interleave of one wave's reads with its own MFMAs stalls on the matrix
// pipe its partner occupies; a burst-ordered form (loads | MFMAs in opposite order on the two wave rows) measured slower still.
//   LDS images per half-tile (8 KiB): LAY_K [128 rows][32 k] (64-byte rows), 16-byte unit u of row r at slot
//   u ^ ((r >> 3) & 3): the four 16-lane groups of a ds_read_b128 fragment read (32 rows x one unit) each touch all 64
//   banks once; LAY_F [32 k][128 rows] — the 256-byte k-rows and rotation of the 64-deep kernels, half as many rows.
// =====================================================================================================
constexpr int kSBK = 32;
constexpr int kSHalfBytes = 8192;

template <int LAY>
struct HOperandS {        // one 1-KiB piece per half-tile and wave (8 waves)
    uint32_t src[2];
    uint64_t base;
    __device__ __forceinline__ void init(const ModeGroup& gFree, int64_t strideK0, uint32_t row0, int wave, int lane) {
        int64_t off[2];
        int64_t mn = INT64_MAX;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (LAY == LAY_K) {
                const int r = 16 * wave + (lane >> 2), p = lane & 3;
                const int u = p ^ ((r >> 3) & 3);
                uint32_t row = row0 + 128 * h + r;
                if (row >= gFree.total) row = gFree.total - 1;   // clamped rows feed outputs that are never stored
                off[h] = (group_offset<0>(gFree, row) + 8 * u) * 2;
            } else {
                const int kk = 4 * wave + (lane >> 4), p = lane & 15;
                const int u = p ^ (4 * ((lane >> 4) & 3));
                uint32_t row = row0 + 128 * h + 8 * u;
                if (row >= gFree.total) row = gFree.total - 8;   // extent % 8 == 0: a unit is all in or all out
                off[h] = (group_offset<0>(gFree, row) + (int64_t)kk * strideK0) * 2;
            }
            mn = off[h] < mn ? off[h] : mn;
        }
        const int64_t mnW = (int64_t)h_uniform64((uint64_t)h_wave_min(mn));
        base = (uint64_t)mnW;
        src[0] = (uint32_t)(off[0] - mnW);
        src[1] = (uint32_t)(off[1] - mnW);
    }
    template <bool PAD>
    __device__ __forceinline__ void issue(HRsrc X, int h, uint32_t slotByte, int wave) const {
        h_dma16<PAD>(X, src[h], slotByte + (uint32_t)wave * 1024u);
    }
};

__device__ __forceinline__ uint32_t h_offK32(int lane, int s) {     // fragment read of k-step s (0, 1) in the 64-byte-row image
    const int u = (lane >> 5) + 2 * s;
    return (uint32_t)((lane & 31) * 64 + ((u ^ ((lane >> 3) & 3)) << 4));
}
template <int LAY>
__device__ __forceinline__ s16x8 h_read_frag32(const char* slot, int rb, int s, const uint32_t (&offK)[2], uint32_t offF) {
    if constexpr (LAY == LAY_K) {
        return *reinterpret_cast<const s16x8*>(slot + rb * 64 + offK[s]);
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef s16x4 __attribute__((address_space(3))) * lptr;
        const char* p = slot + s * 4096 + offF;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p + 1024));
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#else
        (void)slot; (void)rb; (void)s; (void)offK; (void)offF; return s16x8{};
#endif
    }
}

// K-tile walk in steps of 32 (HOdometer with the tile length as a member)
struct HOdometerS {
    uint32_t j0, n0, j1, e1, hi;
    uint64_t offA, offB, stepA, stepB, wrapA, wrapB;
    __device__ __forceinline__ void init(const ModeGroup& gK, uint32_t k0) {
        const uint32_t E0 = gK.div[0].d;
        n0 = HOdometer::sgpr(E0 / kSBK);
        e1 = HOdometer::sgpr(gK.div[1].d);
        const uint32_t q0 = (E0 < 2) ? k0 : fast_div(k0, gK.div[0]);
        j0 = HOdometer::sgpr((k0 - q0 * E0) / kSBK);
        hi = HOdometer::sgpr((e1 < 2) ? q0 : fast_div(q0, gK.div[1]));
        j1 = HOdometer::sgpr(q0 - hi * e1);
        offA = h_uniform64((uint64_t)(group_offset<0>(gK, k0) * 2));
        offB = h_uniform64((uint64_t)(group_offset<1>(gK, k0) * 2));
        stepA = h_uniform64((uint64_t)((int64_t)kSBK * gK.stride[0][0] * 2));
        stepB = h_uniform64((uint64_t)((int64_t)kSBK * gK.stride[1][0] * 2));
        wrapA = h_uniform64((uint64_t)(gK.stride[0][1] * 2) - (uint64_t)(n0 - 1) * stepA);
        wrapB = h_uniform64((uint64_t)(gK.stride[1][1] * 2) - (uint64_t)(n0 - 1) * stepB);
    }
    __device__ __forceinline__ void advance(const ModeGroup& gK) {
        const bool c0 = (j0 + 1 == n0);
        j0 = c0 ? 0u : j0 + 1;
        offA += c0 ? wrapA : stepA;
        offB += c0 ? wrapB : stepB;
        j1 += c0 ? 1u : 0u;
        if (j1 == e1) {
            j1 = 0;
            hi += 1;
            const uint32_t k = hi * e1 * gK.div[0].d;
            if (k < gK.total) {
                offA = h_uniform64((uint64_t)(group_offset<0>(gK, k) * 2));
                offB = h_uniform64((uint64_t)(group_offset<1>(gK, k) * 2));
            }
        }
    }
};

template <bool BF, int LA, int LB, int NS>
__global__ void __launch_bounds__(512, 2) gett_h16s_kernel(const GettParams p) {
    __shared__ __attribute__((aligned(16))) char lds[NS * 4 * kSHalfBytes];
    prefetch_kernarg<(int)sizeof(GettParams)>();
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

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
    const uint32_t kTilesAll = p.gK.total / kSBK, tilesPerSlice = p.kPerSlice / kSBK;
    const uint32_t tile0 = slice * tilesPerSlice;
    const int nTiles = (int)((tile0 + tilesPerSlice <= kTilesAll) ? tilesPerSlice : (kTilesAll - tile0));

    uint64_t bA = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)));
    uint64_t bB = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)));
    HOperandS<LA> oa;
    HOperandS<LB> ob;
    oa.init(p.gM, p.gK.stride[0][0], m0, wave, lane);
    ob.init(p.gN, p.gK.stride[1][0], n0, wave, lane);
    bA += oa.base;
    bB += ob.base;

    uint32_t offK[2], offFa[4], offFb[2];
    offK[0] = h_offK32(lane, 0);
    offK[1] = h_offK32(lane, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) offFa[i] = h_offF(lane, i);
    offFb[0] = h_offF(lane, 2 * (wc & 1));
    offFb[1] = h_offF(lane, 2 * (wc & 1) + 1);

    HOdometerS odo;
    odo.init(p.gK, tile0 * kSBK);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // the wave's four pieces of the K-tile the odometer describes, into ring buffer P: A-half 0, A-half 1, B-half 0, B-half 1
#define CTAMD_S_DMA(P, Q, PAD)                                                                                      \
    {                                                                                                              \
        if constexpr ((Q) < 2) oa.template issue<PAD>(h_make_rsrc(bA + odo.offA), (Q), ldsBase + ((P) * 4 + (Q)) * kSHalfBytes, wave);      \
        else ob.template issue<PAD>(h_make_rsrc(bB + odo.offB), (Q) - 2, ldsBase + ((P) * 4 + (Q)) * kSHalfBytes, wave);                 \
    }
    // ---- prologue: fill the whole ring (tiles 0 .. NS - 1; past the end the last tile is re-staged, never read) ----
    int tNext = 0;                                // K-tile the odometer describes
#define CTAMD_S_FILL(P)                                                                                             \
    if constexpr ((P) < NS) {                                                                                      \
        CTAMD_S_DMA((P) < NS ? (P) : 0, 0, true) CTAMD_S_DMA((P) < NS ? (P) : 0, 1, true)                          \
        CTAMD_S_DMA((P) < NS ? (P) : 0, 2, true) CTAMD_S_DMA((P) < NS ? (P) : 0, 3, true)                          \
        ++tNext;                                                                                                   \
        if (tNext < nTiles) odo.advance(p.gK);                                                                     \
    }
    CTAMD_S_FILL(0) CTAMD_S_FILL(1) CTAMD_S_FILL(2) CTAMD_S_FILL(3) CTAMD_S_FILL(4) CTAMD_S_FILL(5)
    CTAMD_H_VMCNT(4 * (NS - 1));                  // this wave's pieces of tile 0
    __builtin_amdgcn_s_barrier();

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    s16x8 a[2][4], b[2][2];                       // two register sets: k-step s uses set s

    const char* const aSlot0 = lds + wr * kSHalfBytes;                 // A-half wr of buffer 0 (buffer P: + 4 P slots)
    const char* const bSlot0 = lds + (2 + (wc >> 1)) * kSHalfBytes;    // B-half wc >> 1 of buffer 0
    const int bRow = 64 * (wc & 1);
#define CTAMD_S_READ(P, S, SET, F)                                                                                  \
    {                                                                                                              \
        if constexpr ((F) < 2) b[SET][F] = h_read_frag32<LB>(bSlot0 + (P) * 4 * kSHalfBytes, bRow + 32 * (F), S, offK, offFb[F]);       \
        else a[SET][(F) - 2] = h_read_frag32<LA>(aSlot0 + (P) * 4 * kSHalfBytes, 32 * ((F) - 2), S, offK, offFa[(F) - 2]);     \
    }
#define CTAMD_S_MFMA(SET, M) acc[(M) >> 1][(M) & 1] = h_mfma<BF>(a[SET][(M) >> 1], b[SET][(M) & 1], acc[(M) >> 1][(M) & 1]);
    // k-step 0 of the tile in buffer P: reads of k-step 1 beside the MFMAs of k-step 0
#define CTAMD_S_G0(P, G)                                                                                            \
    if constexpr ((G) == 0) { CTAMD_S_READ(P, 1, 1, 0) CTAMD_S_READ(P, 1, 1, 1) }                                   \
    if constexpr ((G) == 1) { CTAMD_S_READ(P, 1, 1, 2) CTAMD_S_READ(P, 1, 1, 3) }                                   \
    if constexpr ((G) == 2) { CTAMD_S_READ(P, 1, 1, 4) }                                                            \
    if constexpr ((G) == 3) { CTAMD_S_READ(P, 1, 1, 5) }                                                            \
    CTAMD_S_MFMA(0, 2 * (G)) CTAMD_S_MFMA(0, 2 * (G) + 1)                                                          \
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1 (after the barrier): reads of the next tile's k-step 0 from buffer PN, one piece of tile t + NS into buffer P
#define CTAMD_S_G1(P, PN, G)                                                                                        \
    if constexpr ((G) == 0) { CTAMD_S_READ(PN, 0, 0, 0) CTAMD_S_READ(PN, 0, 0, 1) }                                 \
    if constexpr ((G) == 1) { CTAMD_S_READ(PN, 0, 0, 2) CTAMD_S_READ(PN, 0, 0, 3) }                                 \
    if constexpr ((G) == 2) { CTAMD_S_READ(PN, 0, 0, 4) }                                                           \
    if constexpr ((G) == 3) { CTAMD_S_READ(PN, 0, 0, 5) }                                                           \
    CTAMD_S_MFMA(1, 2 * (G))                                                                                       \
    CTAMD_S_DMA(P, G, false)                                                                                       \
    CTAMD_S_MFMA(1, 2 * (G) + 1)                                                                                   \
    __builtin_amdgcn_sched_barrier(0);
#define CTAMD_S_SYNC()                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        CTAMD_H_LGKM0();                                                                                           \
        CTAMD_H_VMCNT(4 * (NS - 2));              /* tile t + 1 has landed (this wave's pieces) */                 \
        __builtin_amdgcn_s_barrier();                                                                              \
        __builtin_amdgcn_sched_barrier(0);
#define CTAMD_S_TILE(P, W)                                                                                          \
    {                                                                                                              \
        constexpr int PN_ = ((P) + 1) % NS;                                                                        \
        CTAMD_S_G0(P, 0) CTAMD_S_G0(P, 1) CTAMD_S_G0(P, 2) CTAMD_S_G0(P, 3)                                        \
        CTAMD_S_SYNC()                                                                                             \
        CTAMD_S_G1(P, PN_, 0) CTAMD_S_G1(P, PN_, 1) CTAMD_S_G1(P, PN_, 2) CTAMD_S_G1(P, PN_, 3)                    \
        ++tNext;                                                                                                   \
        if (tNext < nTiles) odo.advance(p.gK);                                                                     \
    }
#define CTAMD_S_LOOP(W)                                                                                             \
    {                                                                                                              \
        int t = 0;                                                                                                 \
        if constexpr (NS == 4) {                                                                                   \
            for (; t + 3 < nTiles; t += 4) { CTAMD_S_TILE(0, W) CTAMD_S_TILE(1, W) CTAMD_S_TILE(2, W) CTAMD_S_TILE(3, W) }   \
            if (t < nTiles) { CTAMD_S_TILE(0, W) }                                                                 \
            if (t + 1 < nTiles) { CTAMD_S_TILE(1, W) }                                                             \
            if (t + 2 < nTiles) { CTAMD_S_TILE(2, W) }                                                             \
        } else {                                                                                                   \
            for (; t + 4 < nTiles; t += 5) { CTAMD_S_TILE(0, W) CTAMD_S_TILE(1, W) CTAMD_S_TILE(2, W) CTAMD_S_TILE(3, W) CTAMD_S_TILE(4, W) } \
            if (t < nTiles) { CTAMD_S_TILE(0, W) }                                                                 \
            if (t + 1 < nTiles) { CTAMD_S_TILE(1, W) }                                                             \
            if (t + 2 < nTiles) { CTAMD_S_TILE(2, W) }                                                             \
            if (t + 3 < nTiles) { CTAMD_S_TILE(3, W) }                                                             \
        }                                                                                                          \
    }

    // first fragments of tile 0
    CTAMD_S_READ(0, 0, 0, 0) CTAMD_S_READ(0, 0, 0, 1) CTAMD_S_READ(0, 0, 0, 2)
    CTAMD_S_READ(0, 0, 0, 3) CTAMD_S_READ(0, 0, 0, 4) CTAMD_S_READ(0, 0, 0, 5)
    CTAMD_S_LOOP(2)
    CTAMD_H_VMCNT(0);                             // the re-staged tail: no LDS-DMA may outlive the workgroup
    h_epilogue_128x64<BF>(p, acc, m0, n0, wr, wc, slice, l, lane, lds, wave);
}

// =====================================================================================================
// Four-wave STREAMED variant (CUTENSOR_AMD_H16_WAVES=4s; round 3, a measured alternative and an autotuning candidate):
// the 128 x 128 wave tile of gett_h16w4_kernel (one wave per SIMD, half the LDS read traffic per MFMA of the 2 x 4
// arrangement) on the K-tile-32 / NS-deep-ring machinery of gett_h16s_kernel.  Why: the four-wave kernel above keeps only
// TWO 64-deep K-tiles in LDS, so a tile's pieces are requested 3-4 k-steps (0.6-0.85 us) before they are needed — less than
// the loaded L2 / fabric latency — and the wave waits at every tile boundary (its LDS-DMA "costs" 21 % of the cycles on
// zero-filled operands, profiles/r03_h16_w4_status.txt, although the issue slots themselves cost ~7 %).  Here a ring of
// NS = 4 or 5 stages of 32 KiB keeps NS - 1 K-tiles (96-128 KiB per CU, 3-4 k-tile times = 1.3-1.7 us on zeros) in flight
// behind a counted vmcnt.  Per 32-deep K-tile a wave issues 32 MFMAs (two k-steps of 16), 16 fragment reads (two register
// sets), 8 LDS-DMA pieces, and meets the workgroup once — in front of k-step 1, whose fragments are already in registers.
// =====================================================================================================
// SPREAD (CUTENSOR_AMD_H16_SPREAD=1, round 3 probe): the eight pieces of a K-tile go out one per FOUR MFMAs over a whole tile time —
// the A pieces (0..3) in k-step 1 of tile t, the B pieces (4..7) in k-step 0 of tile t + 1 — instead of one per two MFMAs
// in k-step 1 only; the deep ring is what allows it (the two-buffer kernel's issue window ends at its landing deadline).
template <bool BF, int LA, int LB, int NS, bool SPREAD = false>
__global__ void __launch_bounds__(256, 1) gett_h16w4s_kernel(const GettParams p) {
    __shared__ __attribute__((aligned(16))) char lds[NS * 4 * kSHalfBytes];
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
    const uint32_t kTilesAll = p.gK.total / kSBK, tilesPerSlice = p.kPerSlice / kSBK;
    const uint32_t tile0 = slice * tilesPerSlice;
    const int nTiles = (int)((tile0 + tilesPerSlice <= kTilesAll) ? tilesPerSlice : (kTilesAll - tile0));

    const uint64_t bA0 = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.A) + group_offset<0>(p.gL, l)));
    const uint64_t bB0 = h_uniform64((uint64_t)(uintptr_t)(static_cast<const uint16_t*>(p.B) + group_offset<1>(p.gL, l)));
    // a half-tile is eight 1-KiB pieces; wave w stages pieces w and w + 4 of each of the four half-tiles
    HOperandS<LA> oaLo, oaHi;
    HOperandS<LB> obLo, obHi;
    oaLo.init(p.gM, p.gK.stride[0][0], m0, wave, lane);
    oaHi.init(p.gM, p.gK.stride[0][0], m0, wave + 4, lane);
    obLo.init(p.gN, p.gK.stride[1][0], n0, wave, lane);
    obHi.init(p.gN, p.gK.stride[1][0], n0, wave + 4, lane);
    const uint64_t bALo = bA0 + oaLo.base, bAHi = bA0 + oaHi.base, bBLo = bB0 + obLo.base, bBHi = bB0 + obHi.base;

    uint32_t offK[2], offF[4];
    offK[0] = h_offK32(lane, 0);
    offK[1] = h_offK32(lane, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) offF[i] = h_offF(lane, i);

    HOdometerS odo;
    odo.init(p.gK, tile0 * kSBK);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // piece n = 0..7 of the K-tile the odometer describes, into ring buffer P: half q = n >> 1 (A0, A1, B0, B1), lo / hi piece
#define CTAMD_Q_DMA(P, N, PAD)                                                                                      \
    {                                                                                                              \
        constexpr int q_ = (N) >> 1;                                                                               \
        constexpr bool hi_ = ((N) & 1) != 0;                                                                       \
        const uint32_t slot_ = ldsBase + ((P) * 4 + q_) * kSHalfBytes;                                             \
        if constexpr (q_ < 2 && !hi_) oaLo.template issue<PAD>(h_make_rsrc(bALo + odo.offA), q_, slot_, wave);     \
        else if constexpr (q_ < 2)    oaHi.template issue<PAD>(h_make_rsrc(bAHi + odo.offA), q_, slot_, wave + 4); \
        else if constexpr (!hi_)      obLo.template issue<PAD>(h_make_rsrc(bBLo + odo.offB), q_ - 2, slot_, wave); \
        else                          obHi.template issue<PAD>(h_make_rsrc(bBHi + odo.offB), q_ - 2, slot_, wave + 4); \
    }
    int tNext = 0;                                // K-tile the odometer describes
    // SPREAD: the last buffer only gets its A pieces here and the odometer stays on that tile (k-step 0 of tile 0 sends the rest)
#define CTAMD_Q_FILL(P)                                                                                             \
    if constexpr ((P) < NS) {                                                                                      \
        CTAMD_Q_DMA((P) < NS ? (P) : 0, 0, true) CTAMD_Q_DMA((P) < NS ? (P) : 0, 1, true)                          \
        CTAMD_Q_DMA((P) < NS ? (P) : 0, 2, true) CTAMD_Q_DMA((P) < NS ? (P) : 0, 3, true)                          \
        if constexpr (!(SPREAD && (P) == NS - 1)) {                                                                \
            CTAMD_Q_DMA((P) < NS ? (P) : 0, 4, true) CTAMD_Q_DMA((P) < NS ? (P) : 0, 5, true)                      \
            CTAMD_Q_DMA((P) < NS ? (P) : 0, 6, true) CTAMD_Q_DMA((P) < NS ? (P) : 0, 7, true)                      \
            ++tNext;                                                                                               \
            if (tNext < nTiles) odo.advance(p.gK);                                                                 \
        }                                                                                                          \
    }
    CTAMD_Q_FILL(0) CTAMD_Q_FILL(1) CTAMD_Q_FILL(2) CTAMD_Q_FILL(3) CTAMD_Q_FILL(4)
    CTAMD_H_VMCNT(8 * (NS - 1) - (SPREAD ? 4 : 0));   // this wave's pieces of tile 0
    __builtin_amdgcn_s_barrier();

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    s16x8 a[2][4], b[2][4];                       // two register sets: k-step s uses set s

    const char* const aSlot0 = lds + wr * kSHalfBytes;          // A-half wr of buffer 0 (buffer P: + 4 P slots)
    const char* const bSlot0 = lds + (2 + wc) * kSHalfBytes;    // B-half wc of buffer 0
    // fragment F = 0..7 of k-step S from buffer P into register set SET: F < 4 -> B columns 32 F, else A rows 32 (F - 4)
#define CTAMD_Q_READ(P, S, SET, F)                                                                                  \
    {                                                                                                              \
        if constexpr ((F) < 4) b[SET][F] = h_read_frag32<LB>(bSlot0 + (P) * 4 * kSHalfBytes, 32 * (F), S, offK, offF[F]);               \
        else a[SET][(F) - 4] = h_read_frag32<LA>(aSlot0 + (P) * 4 * kSHalfBytes, 32 * ((F) - 4), S, offK, offF[(F) - 4]);               \
    }
#define CTAMD_Q_MFMA(SET, M) acc[(M) >> 2][(M) & 3] = h_mfma<BF>(a[SET][(M) >> 2], b[SET][(M) & 3], acc[(M) >> 2][(M) & 3]);
    // k-step 0 of the tile in buffer P: one fragment read of k-step 1 per two MFMAs of k-step 0
#define CTAMD_Q_G0(P, G)                                                                                            \
    CTAMD_Q_READ(P, 1, 1, G)                                                                                       \
    CTAMD_Q_MFMA(0, 2 * (G))                                                                                       \
    if constexpr (SPREAD && ((G) & 1)) CTAMD_Q_DMA(((P) + NS - 1) % NS, 4 + ((G) >> 1), false)                     \
    CTAMD_Q_MFMA(0, 2 * (G) + 1)                                                                                   \
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1 (behind the barrier): one read of the next tile's k-step 0 (buffer PN) and one piece of tile t + NS (into buffer
    // P, which nobody reads any more) per two MFMAs
#define CTAMD_Q_G1(P, PN, G)                                                                                        \
    CTAMD_Q_READ(PN, 0, 0, G)                                                                                      \
    CTAMD_Q_MFMA(1, 2 * (G))                                                                                       \
    if constexpr (!SPREAD) CTAMD_Q_DMA(P, G, false)                                                                \
    else if constexpr (((G) & 1) == 0) CTAMD_Q_DMA(P, (G) >> 1, false)                                             \
    CTAMD_Q_MFMA(1, 2 * (G) + 1)                                                                                   \
    __builtin_amdgcn_sched_barrier(0);
#define CTAMD_Q_SYNC()                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        CTAMD_H_LGKM0();                                                                                           \
        CTAMD_H_VMCNT(8 * (NS - 2));              /* tile t + 1 has landed (this wave's pieces) */                 \
        __builtin_amdgcn_s_barrier();                                                                              \
        __builtin_amdgcn_sched_barrier(0);
#define CTAMD_Q_TILE(P)                                                                                             \
    {                                                                                                              \
        constexpr int PN_ = ((P) + 1) % NS;                                                                        \
        CTAMD_Q_G0(P, 0) CTAMD_Q_G0(P, 1) CTAMD_Q_G0(P, 2) CTAMD_Q_G0(P, 3)                                        \
        CTAMD_Q_G0(P, 4) CTAMD_Q_G0(P, 5) CTAMD_Q_G0(P, 6) CTAMD_Q_G0(P, 7)                                        \
        if constexpr (SPREAD) { ++tNext; if (tNext < nTiles) odo.advance(p.gK); }                                  \
        CTAMD_Q_SYNC()                                                                                             \
        CTAMD_Q_G1(P, PN_, 0) CTAMD_Q_G1(P, PN_, 1) CTAMD_Q_G1(P, PN_, 2) CTAMD_Q_G1(P, PN_, 3)                    \
        CTAMD_Q_G1(P, PN_, 4) CTAMD_Q_G1(P, PN_, 5) CTAMD_Q_G1(P, PN_, 6) CTAMD_Q_G1(P, PN_, 7)                    \
        if constexpr (!SPREAD) { ++tNext; if (tNext < nTiles) odo.advance(p.gK); }                                 \
    }

    // first fragments of tile 0
    CTAMD_Q_READ(0, 0, 0, 0) CTAMD_Q_READ(0, 0, 0, 1) CTAMD_Q_READ(0, 0, 0, 2) CTAMD_Q_READ(0, 0, 0, 3)
    CTAMD_Q_READ(0, 0, 0, 4) CTAMD_Q_READ(0, 0, 0, 5) CTAMD_Q_READ(0, 0, 0, 6) CTAMD_Q_READ(0, 0, 0, 7)
    {
        int t = 0;
        if constexpr (NS == 4) {
            for (; t + 3 < nTiles; t += 4) { CTAMD_Q_TILE(0) CTAMD_Q_TILE(1) CTAMD_Q_TILE(2) CTAMD_Q_TILE(3) }
            if (t < nTiles) { CTAMD_Q_TILE(0) }
            if (t + 1 < nTiles) { CTAMD_Q_TILE(1) }
            if (t + 2 < nTiles) { CTAMD_Q_TILE(2) }
        } else {
            for (; t + 4 < nTiles; t += 5) { CTAMD_Q_TILE(0) CTAMD_Q_TILE(1) CTAMD_Q_TILE(2) CTAMD_Q_TILE(3) CTAMD_Q_TILE(4) }
            if (t < nTiles) { CTAMD_Q_TILE(0) }
            if (t + 1 < nTiles) { CTAMD_Q_TILE(1) }
            if (t + 2 < nTiles) { CTAMD_Q_TILE(2) }
            if (t + 3 < nTiles) { CTAMD_Q_TILE(3) }
        }
    }
    CTAMD_H_VMCNT(0);                             // the re-staged tail: no LDS-DMA may outlive the workgroup

    const uint32_t mW = m0 + 128 * wr, nW = n0 + 128 * wc;    // this wave's quadrant
    if (p.partial != nullptr) {                   // split-K: fp32 partial tile, row-major [slice][l][m][n]
        const uint32_t Mt = p.gM.total, Nt = p.gN.total;
        float* P = p.partial + ((size_t)slice * p.gL.total + l) * (size_t)Mt * Nt;
        auto store_partial = [&](const f32x16& c0, const f32x16& c1, const f32x16& c2, const f32x16& c3, uint32_t mBase) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t m = mBase + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < Mt) {
                    const uint32_t n = nW + (lane & 31);
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
        ep.park(0, acc[i][0], lane); ep.park(1, acc[i][1], lane); ep.park(2, acc[i][2], lane); ep.park(3, acc[i][3], lane);
        const uint32_t mB = mW + 32 * i;
        ep.template flush<BF>(p, mB, 0u, 0u, nW, 64u, 32u, lane);
    }
}

template <bool BF, int LA, int LB>
static hipError_t launch_h16w4s(const GettParams& p, hipStream_t stream) {
    static const int ns = [] { const char* e = getenv("CUTENSOR_AMD_H16_STAGES"); return e ? atoi(e) : 5; }();
    static const bool spread = [] { const char* e = getenv("CUTENSOR_AMD_H16_SPREAD"); return e && e[0] == '1'; }();
    if (spread) hipLaunchKernelGGL((gett_h16w4s_kernel<BF, LA, LB, 4, true>), dim3(p.nBlocks), dim3(256), 0, stream, p);
    else if (ns == 4) hipLaunchKernelGGL((gett_h16w4s_kernel<BF, LA, LB, 4>), dim3(p.nBlocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gett_h16w4s_kernel<BF, LA, LB, 5>), dim3(p.nBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <bool BF, int LA, int LB>
static hipError_t launch_h16s(const GettParams& p, hipStream_t stream) {
    static const int ns = [] { const char* e = getenv("CUTENSOR_AMD_H16_STAGES"); return e ? atoi(e) : 5; }();
    if (ns == 4) hipLaunchKernelGGL((gett_h16s_kernel<BF, LA, LB, 4>), dim3(p.nBlocks), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((gett_h16s_kernel<BF, LA, LB, 5>), dim3(p.nBlocks), dim3(512), 0, stream, p);
    return hipGetLastError();
}

template <bool BF, int LA, int LB>
static hipError_t launch_h16w4(const GettParams& p, hipStream_t stream) {
    static const int abl = [] { const char* e = getenv("CUTENSOR_AMD_H16_ABL"); return e ? atoi(e) : 0; }();
    if constexpr (BF && LA == LAY_K && LB == LAY_K) {
        if (abl == 1) { hipLaunchKernelGGL((gett_h16w4_kernel<BF, LA, LB, 1>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (abl == 2) { hipLaunchKernelGGL((gett_h16w4_kernel<BF, LA, LB, 2>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (abl == 3) { hipLaunchKernelGGL((gett_h16w4_kernel<BF, LA, LB, 3>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (abl == 4) { hipLaunchKernelGGL((gett_h16w4_kernel<BF, LA, LB, 4>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
        if (abl == 5) { hipLaunchKernelGGL((gett_h16w4_kernel<BF, LA, LB, 5>), dim3(p.nBlocks), dim3(256), 0, stream, p); return hipGetLastError(); }
    }
    hipLaunchKernelGGL((gett_h16w4_kernel<BF, LA, LB>), dim3(p.nBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <bool BF, int LA, int LB>
static hipError_t launch_h16(const GettParams& p, hipStream_t stream) {
    static const bool timed = [] { const char* e = getenv("CUTENSOR_AMD_H16_TIMED"); return e && e[0] == '1'; }();
    const dim3 grid(p.nBlocks);
    if constexpr (BF && LA == LAY_K && LB == LAY_F) {   // the one instantiation that carries the in-kernel timestamps
        if (timed) {
            hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB, true>), grid, dim3(512), 0, stream, p);
            return hipGetLastError();
        }
    }
    if constexpr (BF && LA == LAY_K && LB == LAY_K) {   // ablations: one instantiation ('km,kn' bf16)
        static const int abl = [] { const char* e = getenv("CUTENSOR_AMD_H16_ABL"); return e ? atoi(e) : 0; }();
        if (abl == 1) { hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB, false, 1>), grid, dim3(512), 0, stream, p); return hipGetLastError(); }
        if (abl == 2) { hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB, false, 2>), grid, dim3(512), 0, stream, p); return hipGetLastError(); }
        if (abl == 3) { hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB, false, 3>), grid, dim3(512), 0, stream, p); return hipGetLastError(); }
        if (abl == 4) { hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB, false, 4>), grid, dim3(512), 0, stream, p); return hipGetLastError(); }
        if (abl == 5) { hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB, false, 5>), grid, dim3(512), 0, stream, p); return hipGetLastError(); }
        if (abl == 6) { hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB, false, 6>), grid, dim3(512), 0, stream, p); return hipGetLastError(); }
        if (abl == 7) { hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB, false, 7>), grid, dim3(512), 0, stream, p); return hipGetLastError(); }
        if (abl == 8) { hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB, false, 8>), grid, dim3(512), 0, stream, p); return hipGetLastError(); }
    }
    hipLaunchKernelGGL((gett_h16_kernel<BF, LA, LB>), grid, dim3(512), 0, stream, p);
    return hipGetLastError();
}

#endif  // CTAMD_RESEARCH_KERNELS

// ---------------------------------------------------------------------------------------------------------------------
// Measurement only: the rate at which THIS device sustains nothing but independent v_mfma_f32_32x32x16_{bf16,f16} on a
// given kind of operand data (one wave per SIMD, eight operand register pairs, no memory traffic at all).  On zeros it
// is the nominal peak; on U(-1,1) data the power limit pulls the clock down and the rate differs from box to box by up to
// 30 % — bench.py quotes the GETT kernel against both.
// ---------------------------------------------------------------------------------------------------------------------
template <bool BF>
__global__ void __launch_bounds__(256, 1) mfma_ceiling_kernel(const s16x8* __restrict__ data, float* out, int iters) {
    s16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = data[(i * 2 + 0) * 256 + threadIdx.x];
        b[i] = data[(i * 2 + 1) * 256 + threadIdx.x];
    }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 7] = h_mfma<BF>(a[i & 7], b[(i >> 1) & 7], acc[i & 7]);
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][7];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// the same for v_mfma_f32_16x16x32_{bf16,f16} (the default 16-bit kernel's instruction, gett_h16v.hip), issued from inline asm like there
template <bool BF>
__global__ void __launch_bounds__(256, 1) mfma16_ceiling_kernel(const s16x8* __restrict__ data, float* out, int iters) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    s16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = data[(i * 2 + 0) * 256 + threadIdx.x];
        b[i] = data[(i * 2 + 1) * 256 + threadIdx.x];
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {   // 32 x (16x16x32) = the flops of 16 x (32x32x16)
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (BF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i & 15]) : "v"(a[i & 7]), "v"(b[(i >> 1) & 7]));
            else              asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i & 15]) : "v"(a[i & 7]), "v"(b[(i >> 1) & 7]));
#endif
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

}  // namespace ctamd

// dataKind 0: zeros, 1: U(-1,1) (fixed seed); shape 0: 32x32x16, 1: 16x16x32.  Returns 0 and the sustained TFLOP/s (after ~40 ms of
// burn-in), or -1.
extern "C" int ctamdMeasureMfmaCeilingShape(int bf16, int dataKind, int shape, float* tflops);
extern "C" int ctamdMeasureMfmaCeiling(int bf16, int dataKind, float* tflops) { return ctamdMeasureMfmaCeilingShape(bf16, dataKind, 0, tflops); }
extern "C" int ctamdMeasureMfmaCeilingShape(int bf16, int dataKind, int shape, float* tflops) {
    using namespace ctamd;
    if (tflops == nullptr) return -1;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    const int cus = prop.multiProcessorCount;
    const size_t n = 16 * 256 * 8;     // 8 operand pairs x 256 lanes x 8 elements
    uint16_t* h = static_cast<uint16_t*>(malloc(n * 2));
    if (h == nullptr) return -1;
    uint32_t lcg = 12345u;
    for (size_t i = 0; i < n; ++i) {
        lcg = lcg * 1664525u + 1013904223u;
        const float x = dataKind == 0 ? 0.f : 2.f * (float)(lcg >> 8) / 16777216.f - 1.f;
        if (bf16) {
            uint32_t u;
            memcpy(&u, &x, 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            h[i] = (uint16_t)(u >> 16);
        } else {
            const _Float16 hf = (_Float16)x;
            memcpy(&h[i], &hf, 2);
        }
    }
    s16x8* d = nullptr;
    float* out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = -1;
    if (hipMalloc((void**)&d, n * 2) == hipSuccess && hipMalloc((void**)&out, (size_t)cus * 256 * 4) == hipSuccess &&
        hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        const int iters = 20000;       // ~12-25 ms per launch
        auto launch = [&]() {
            if (shape == 1) {
                if (bf16) hipLaunchKernelGGL((mfma16_ceiling_kernel<true>), dim3(cus), dim3(256), 0, nullptr, d, out, iters);
                else      hipLaunchKernelGGL((mfma16_ceiling_kernel<false>), dim3(cus), dim3(256), 0, nullptr, d, out, iters);
            } else {
                if (bf16) hipLaunchKernelGGL((mfma_ceiling_kernel<true>), dim3(cus), dim3(256), 0, nullptr, d, out, iters);
                else      hipLaunchKernelGGL((mfma_ceiling_kernel<false>), dim3(cus), dim3(256), 0, nullptr, d, out, iters);
            }
        };
        for (int w = 0; w < 3; ++w) launch();
        (void)hipEventRecord(e0, nullptr);
        for (int w = 0; w < 3; ++w) launch();
        (void)hipEventRecord(e1, nullptr);
        float ms = 0.f;
        if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f) {
            const double flops = 3.0 * (double)cus * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
            *tflops = (float)(flops / (ms * 1e-3) / 1e12);
            rc = 0;
        }
    }
    (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d) (void)hipFree(d);
    if (out) (void)hipFree(out);
    free(h);
    return rc;
}

namespace ctamd {

// bf16 entries first, then fp16, each in the order (layA, layB) = (K,K) (K,F) (F,K) (F,F)
#if !defined(CTAMD_RESEARCH_KERNELS)
static hipError_t launch_h16_not_built(const GettParams&, hipStream_t) { return hipErrorNotSupported; }
#define CTAMD_H16_ENTRY(bf, la, lb)    {kHTile, kHTile, kHBK, 2, 4, 1, la, lb, 512, 5, 1, 2, &launch_h16_not_built, 0},
#define CTAMD_H16W4_ENTRY(bf, la, lb)  {kHTile, kHTile, kHBK, 2, 2, 1, la, lb, 256, 2, 1, 2, &launch_h16_not_built, 0},
#define CTAMD_H16S_ENTRY(bf, la, lb)   {kHTile, kHTile, 32, 2, 4, 1, la, lb, 512, 4, 1, 2, &launch_h16_not_built, 0},
#define CTAMD_H16W4R_ENTRY(bf, la, lb) {kHTile, kHTile, kHBK, 2, 2, 1, la, lb, 256, 3, 1, 2, &launch_h16_not_built, 0},
#define CTAMD_H16W4S_ENTRY(bf, la, lb) {kHTile, kHTile, 32, 2, 2, 1, la, lb, 256, 5, 1, 2, &launch_h16_not_built, 0},
#else
#define CTAMD_H16_ENTRY(bf, la, lb) \
    {kHTile, kHTile, kHBK, 2, 4, 1, la, lb, 512, 5, 1, 0, &launch_h16<bf, la, lb>, 0},
#define CTAMD_H16W4_ENTRY(bf, la, lb) \
    {kHTile, kHTile, kHBK, 2, 2, 1, la, lb, 256, 2, 1, 0, &launch_h16w4<bf, la, lb>, 0},
#define CTAMD_H16S_ENTRY(bf, la, lb) \
    {kHTile, kHTile, kSBK, 2, 4, 1, la, lb, 512, 4, 1, 0, &launch_h16s<bf, la, lb>, 0},
#define CTAMD_H16W4R_ENTRY(bf, la, lb) \
    {kHTile, kHTile, kHBK, 2, 2, 1, la, lb, 256, 3, 1, 0, &launch_h16w4r<bf, la, lb>, 0},
#define CTAMD_H16W4S_ENTRY(bf, la, lb) \
    {kHTile, kHTile, kSBK, 2, 2, 1, la, lb, 256, 5, 1, 0, &launch_h16w4s<bf, la, lb>, 0},
#endif
static const GettKernelInfo g_h16_table[] = {
    CTAMD_H16_ENTRY(true, LAY_K, LAY_K) CTAMD_H16_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16_ENTRY(true, LAY_F, LAY_K) CTAMD_H16_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16_ENTRY(false, LAY_K, LAY_K) CTAMD_H16_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16_ENTRY(false, LAY_F, LAY_K) CTAMD_H16_ENTRY(false, LAY_F, LAY_F)
    // entries 8..15: the four-wave variant, same order
    CTAMD_H16W4_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W4_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W4_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W4_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W4_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W4_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W4_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W4_ENTRY(false, LAY_F, LAY_F)
    // entries 16..23: the streamed eight-wave variant (free-running waves, K-tile 32, deep LDS ring), same order
    CTAMD_H16S_ENTRY(true, LAY_K, LAY_K) CTAMD_H16S_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16S_ENTRY(true, LAY_F, LAY_K) CTAMD_H16S_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16S_ENTRY(false, LAY_K, LAY_K) CTAMD_H16S_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16S_ENTRY(false, LAY_F, LAY_K) CTAMD_H16S_ENTRY(false, LAY_F, LAY_F)
    // entries 24..31: the four-wave streamed variant (128 x 128 wave tiles on the K-tile-32 ring), same order
    CTAMD_H16W4S_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W4S_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W4S_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W4S_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W4S_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W4S_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W4S_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W4S_ENTRY(false, LAY_F, LAY_F)
    // entries 32..39: the four-wave register-staged variant (no LDS-DMA), same order
    CTAMD_H16W4R_ENTRY(true, LAY_K, LAY_K) CTAMD_H16W4R_ENTRY(true, LAY_K, LAY_F)
    CTAMD_H16W4R_ENTRY(true, LAY_F, LAY_K) CTAMD_H16W4R_ENTRY(true, LAY_F, LAY_F)
    CTAMD_H16W4R_ENTRY(false, LAY_K, LAY_K) CTAMD_H16W4R_ENTRY(false, LAY_K, LAY_F)
    CTAMD_H16W4R_ENTRY(false, LAY_F, LAY_K) CTAMD_H16W4R_ENTRY(false, LAY_F, LAY_F)};

// entries 40..47: the four-wave kernel with the lean instruction stream (gett_h16v.hip), 48..55: its 16x16x32 form, 56..63: the
// 128 x 128 mid-size sibling of that (two workgroups per CU), 64..71 / 72..79: that tile on a four-deep ring / with dedicated
// data-moving waves, 80..87: the 64 x 64 tile, 88..95: the persistent 256 x 256 kernel (gett_h16p.hip, round 5); same order
const GettKernelInfo* gett_h16_kernels(int* count) {
    constexpr int nHere = (int)(sizeof(g_h16_table) / sizeof(g_h16_table[0]));
    struct All { GettKernelInfo e[nHere + 56 + 8]; int n; };
    static const All all = [] {
        All a{};
        for (int i = 0; i < nHere; ++i) a.e[i] = g_h16_table[i];
        int nv = 0;
        const GettKernelInfo* v = gett_h16v_kernels(&nv);
        a.n = nHere;
        for (int i = 0; i < nv && i < 56; ++i) a.e[a.n++] = v[i];
        const GettKernelInfo* pk = gett_h16p_kernels(&nv);     // 88..95: the persistent 256 x 256 kernel (gett_h16p.hip)
        for (int i = 0; i < nv && i < 8; ++i) a.e[a.n++] = pk[i];
        return a;
    }();
    *count = all.n;
    return all.e;
}

}  // namespace ctamd
