// gett_f32_stream.hip — fp32 GETT "streaming" kernels for gfx950 (MI355X, CDNA4).
//
// Same contraction semantics and GEMM view as gett_f32.hip (reference call sites:
// cuTENSOR/contraction.cu:261-265, cuTENSOR/einsum.cu:334-338), different machine mapping, built for
// the problems whose time is the K loop itself — above all the headline einsum 'abcd,dcbe->ae'
// (one 96 x 96 output tile, K = 262,144 split over all 256 CUs), where a CU has to sustain
// ~10 B/clk of HBM reads *and* ~100 % MFMA issue at the same time.
//
//   * HBM -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass, no
//     VALU work per byte.  A wave-instruction moves 64 x 16 B into one contiguous 1-KiB piece of LDS;
//     which 16-byte unit of the tile a lane fetches is free, so the LDS image is shaped by permuting
//     the *source* units:
//       - K-contiguous operand (LAY_K): image [row][32 k] (128-B rows, one full HBM line per row);
//         unit p of row r holds k-unit p ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment
//         reads (16 rows x 4 k-units per lane group) hit 16 distinct 16-B slots: conflict-free.
//       - free-contiguous operand (LAY_F): image [32 k][ROWS] (ROWS % 32 == 0); k-rows with bit 2
//         of k set are rotated by 16 floats, so the two k-slots a half-wave reads with ds_read_b32
//         fall in opposite bank halves: conflict-free.
//   * S-deep LDS ring (24 KiB per stage at 96 x 96 x 32): S - 1 K-tiles are in flight per CU, which
//     covers the ~2 us loaded HBM latency; the only wait in the loop is a *counted* vmcnt.
//   * one 8-wave workgroup per CU: four multiplying waves (one per SIMD) and four data-moving waves
//     (one per SIMD) that do nothing but issue LDS-DMA and keep the K-tile odometer — an LDS-DMA
//     instruction blocks its wave's issue for 60-100+ cycles, which on a multiplying wave would drain
//     the one-deep MFMA queue.  The MFMA operand fragments are double buffered in registers and
//     fetched one 16-step ahead, interleaved with the MFMAs of the current step; the per-tile barrier
//     sits a third into the tile's last 16-step, so the fragment reads of the next tile are covered by
//     the MFMAs that are still to be issued.  The ring walk is unrolled S times: every LDS address is
//     a lane-constant base plus an immediate.
//   * split-K partial tiles are written in the accumulator's own register order (16 B per lane,
//     1 KiB per wave-instruction) and folded by splitk_reduce_frag_kernel.
//
// Roofline: fp32 MFMA (v_mfma_f32_16x16x4_f32, 256 flop/clk/CU); algorithmic flops = 2*L*M*N*K,
// algorithmic bytes = |A| + |B| + |D|.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "params.h"
#include "launch.h"
#include "gett_common.h"

namespace ctamd {

constexpr int kStreamBK = 32;

// 64 lanes x 16 B into one contiguous 1-KiB piece of LDS starting at the wave-uniform address dst (the
// compiler moves it to M0).  Source = buffer descriptor (SGPRs) + per-lane byte offset (a loop-invariant
// VGPR) + wave-uniform byte offset of the K-tile (an SGPR): `buffer_load_dwordx4 v, s[0:3], s offen lds`.
// No vector ALU instruction is needed per piece — on gfx950 the fp32 MFMA issues through the same port
// as the vector ALU, so a data-moving wave that needs a v_add per address only gets to issue while the
// multiplying wave of its SIMD is stalled (measured: 290 cycles of barrier wait per K-tile).
// The builtins exist only in the device pass; hipcc's host pass silently drops the stubs of kernel
// templates that mention them, hence the guards.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t BufRsrc;
#else
typedef int BufRsrc;
#endif
__device__ __forceinline__ BufRsrc make_rsrc(const float* base, uint32_t records = 0xffffffffu) {
#if defined(__HIP_DEVICE_COMPILE__)
    // raw buffer, stride 0, num_records = 2^32 - 1 bytes (the planner only selects these kernels for
    // operands whose byte span fits 32 bits), dword 3 = gfx9 raw-buffer format word.  RAG instantiations pass the EXACT
    // number of bytes from `base` to the end of the tensor: the range check is per dword and counts soffset + voffset
    // (tools/ubench/ldsdma_unaligned.hip), so a 16-byte unit that reaches past the last element loads the floats that
    // exist and zeros for the rest — and a lane offset with bit 31 set (spans below 2^31) reads nothing at all.
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)records, 0x00020000);
#else
    (void)base; (void)records; return 0;
#endif
}
__device__ __forceinline__ void store_wt_16(f32x4 v, BufRsrc rsrc, uint32_t byteOff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)byteOff, 0, /*aux: sc1*/ 16);
#else
    (void)v; (void)rsrc; (void)byteOff;
#endif
}
template <int AUX>   // measurement: the same store with another cache policy (0 = plain / write-back, 2 = nontemporal)
__device__ __forceinline__ void store_policy_16(f32x4 v, BufRsrc rsrc, uint32_t byteOff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, (int)byteOff, 0, AUX);
#else
    (void)v; (void)rsrc; (void)byteOff;
#endif
}
template <int AUX = 0>   // AUX = 2: nontemporal ("nt") — streamed once, not worth a line of the Infinity Cache
__device__ __forceinline__ void lds_dma_16(BufRsrc rsrc, uint32_t laneBytes, uint32_t tileBytes, float* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, (int)laneBytes, (int)tileBytes, 0, AUX);
#else
    (void)rsrc; (void)laneBytes; (void)tileBytes; (void)dst;
#endif
}

// ---------------------------------------------------------------------------------------------
// One operand of the streamed tile: where each lane's LDS-DMA pieces come from, and how MFMA operand
// fragments are read back.  ROWS = BM or BN (multiple of 32); 4 waves share the ROWS/8 pieces.
// ---------------------------------------------------------------------------------------------
template <int LAY, int ROWS>
struct StreamOperand {
    static_assert(ROWS % 32 == 0, "tile rows must be a multiple of 32");
    static constexpr int PIECES   = ROWS / 8;        // 1-KiB pieces per tile (either layout)
    static constexpr int PER_WAVE = PIECES / 4;
    static constexpr int FLOATS   = ROWS * kStreamBK;
    static constexpr int UR       = ROWS / 4;        // 16-byte units per k-row (LAY_F)

    uint32_t src[PER_WAVE];  // byte offset of this lane's unit of piece i (tile k0 = 0)

    // SLOT_R / SLOT_K: slot of this tensor in its free group / in the K group
    template <int SLOT_K>
    __device__ __forceinline__ void init(const ModeGroup& gFree, const ModeGroup& gK, uint32_t row0, int wave, int lane) {
        const int64_t strideK0 = gK.stride[SLOT_K][0];
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int c = wave + 4 * i;
            if constexpr (LAY == LAY_K) {
                const int r = 8 * c + (lane >> 3), p = lane & 7;
                const int u = p ^ ((r >> 1) & 7);
                uint32_t row = row0 + r;
                if (row >= gFree.total) row = gFree.total - 1;   // clamped rows feed outputs that are never stored
                src[i] = (group_offset32<0>(gFree, row) + 4u * (uint32_t)u) * 4u;    // K-contiguous: strideK0 == 1
            } else {
                const int g = 64 * c + lane;
                const int kr = g / UR, p = g % UR;
                const int u = (p + 4 * ((kr >> 2) & 1)) % UR;
                uint32_t row = row0 + 4 * u;
                if (row >= gFree.total) row = (gFree.total - 1u) & ~3u;   // the last unit that holds rows of the mode (whole when extent % 4 == 0)
                src[i] = (group_offset32<0>(gFree, row) + (uint32_t)kr * (uint32_t)strideK0) * 4u;
            }
        }
    }

    // issue this wave's pieces of one tile: X + src + offK bytes (wave-uniform) -> stage + piece
    template <int AUX = 0>
    __device__ __forceinline__ void issue(BufRsrc X, uint32_t offKBytes, float* stage, int wave) const {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i)
            lds_dma_16<AUX>(X, src[i], offKBytes, stage + (wave + 4 * i) * 256);
    }

    // Ragged K / operands without 16-byte lanes (RAG instantiations; round 6).  The LAST K-tile of the last slice holds kValid < 32
    // k, or K-contiguous rows whose last 16-byte unit is partial (K % 4 != 0).  mask(): units that lie entirely past the contracted
    // range get bit 31 in their byte offset — out of range for the RAG descriptor (2^31 > records): no memory access, zeros in LDS.
    // Called once, right before that tile is issued (nothing is issued after it).
    __device__ __forceinline__ void mask(int wave, int lane, uint32_t kValid) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int c = wave + 4 * i;
            bool out;
            if constexpr (LAY == LAY_K) {
                const int r = 8 * c + (lane >> 3), p = lane & 7;
                out = 4u * (uint32_t)(p ^ ((r >> 1) & 7)) >= kValid;
            } else {
                out = (uint32_t)((64 * c + lane) / UR) >= kValid;
            }
            src[i] |= out ? 0x80000000u : 0u;
        }
    }
    // fix(): K-contiguous operand, K % 4 != 0 — the partial unit of every row was staged whole (its tail is the head of the next
    // row, or zeros past the end of the tensor): zero floats kValid - 4 u .. 3 in LDS.  Called by the data-moving wave that staged the
    // pieces, behind its own wait for them and in front of the tile's barrier (the caller drains lgkmcnt).
    __device__ __forceinline__ void fix(float* stage, int wave, int lane, uint32_t kValid) const {
        if constexpr (LAY == LAY_K) {
#pragma unroll
            for (int i = 0; i < PER_WAVE; ++i) {
                const int c = wave + 4 * i;
                const int r = 8 * c + (lane >> 3), p = lane & 7;
                const uint32_t k0 = 4u * (uint32_t)(p ^ ((r >> 1) & 7));
                float* at = stage + c * 256 + lane * 4;
                if (k0 < kValid && kValid < k0 + 4u) {
                    const uint32_t v = kValid - k0;           // 1 .. 3 live floats
                    if (v <= 1u) at[1] = 0.f;
                    if (v <= 2u) at[2] = 0.f;
                    at[3] = 0.f;
                }
            }
        }
    }

    // per-lane constant part of the fragment address (floats) for the 16-row fragment at rbase
    __device__ static __forceinline__ int frag_base(int rbase, int lane) {
        const int i = lane & 15, q = lane >> 4;
        if constexpr (LAY == LAY_K) {
            const int r = rbase + i;
            return r * 32 + ((q ^ ((r >> 1) & 7)) << 2);          // 16-step 1 flips bit 2 of the unit: ^ 16 floats
        } else {
            const int n = rbase + i;
            const int p = ((n >> 2) - 4 * (q & 1) + UR) % UR;
            return (4 * q) * ROWS + p * 4 + (n & 3);              // + (16 s + kk) * ROWS
        }
    }

    // operand registers of the four MFMAs of 16-step s: out[kk] feeds MFMA kk (k = 16 s + 4 q + kk)
    template <int STEP>
    __device__ static __forceinline__ f32x4 fragment(const float* lds, int base) {
        if constexpr (LAY == LAY_K) {
            return *reinterpret_cast<const f32x4*>(lds + (STEP ? (base ^ 16) : base));
        } else {
            f32x4 o;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) o[kk] = lds[base + (16 * STEP + kk) * ROWS];
            return o;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Wave-uniform walk over the K-tiles of a slice, one tile (kStreamBK k's) per advance(): element
// offsets of the tile in A and B without a division on the common path.  The fastest contracted digit
// holds n0 tiles per period (fast-K: its extent is a multiple of kStreamBK); stepping inside it and
// the carry into the second digit are additions of precomputed constants; a carry beyond the second
// digit (rare) re-derives the offsets from the linear index.
// ---------------------------------------------------------------------------------------------
struct KOdometer {
    // everything here is wave-uniform and lives in SGPRs (readfirstlane pins it): the walk costs scalar
    // instructions only.  Offsets are bytes modulo 2^32 (operand byte spans fit 32 bits).
    uint32_t j0, n0, j1, e1, hi;
    uint32_t offA, offB, stepA, stepB, wrapA, wrapB;

    __device__ static __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

    // RAG: the (single) contracted mode ends inside its last K-tile — that tile counts
    template <bool RAG = false>
    __device__ __forceinline__ void init(const ModeGroup& gK, uint32_t k0) {
        const uint32_t E0 = gK.div[0].d;
        n0 = sgpr((E0 + (RAG ? (uint32_t)kStreamBK - 1u : 0u)) / kStreamBK);
        e1 = sgpr(gK.div[1].d);
        const uint32_t q0 = (E0 < 2) ? k0 : fast_div(k0, gK.div[0]);
        j0 = sgpr((k0 - q0 * E0) / kStreamBK);
        hi = sgpr((e1 < 2) ? q0 : fast_div(q0, gK.div[1]));
        j1 = sgpr(q0 - hi * e1);
        offA = sgpr((uint32_t)group_offset<0>(gK, k0) * 4u);
        offB = sgpr((uint32_t)group_offset<1>(gK, k0) * 4u);
        stepA = sgpr((uint32_t)((int64_t)kStreamBK * gK.stride[0][0]) * 4u);
        stepB = sgpr((uint32_t)((int64_t)kStreamBK * gK.stride[1][0]) * 4u);
        wrapA = sgpr((uint32_t)gK.stride[0][1] * 4u - (n0 - 1) * stepA);
        wrapB = sgpr((uint32_t)gK.stride[1][1] * 4u - (n0 - 1) * stepB);
    }
    __device__ __forceinline__ void advance(const ModeGroup& gK) {
        const bool c0 = (j0 + 1 == n0);
        j0 = c0 ? 0u : j0 + 1;
        offA += c0 ? wrapA : stepA;
        offB += c0 ? wrapB : stepB;
        j1 += c0 ? 1u : 0u;
        if (j1 == e1) {   // carry beyond the second digit
            j1 = 0;
            hi += 1;
            const uint32_t k = hi * e1 * gK.div[0].d;
            if (k < gK.total) {
                offA = sgpr((uint32_t)group_offset<0>(gK, k) * 4u);
                offB = sgpr((uint32_t)group_offset<1>(gK, k) * 4u);
            }
        }
    }
};

template <int BM_, int BN_, int LA_, int LB_, int S_, int ABL_ = 0, bool RAG_ = false>
struct StreamCfg {
    static constexpr int BM = BM_, BN = BN_, LA = LA_, LB = LB_, S = S_;
    static constexpr bool RAG = RAG_;  // ragged K (one contracted mode, K % 32 != 0) / operands without 16-byte lanes: the last K-tile of the
                                       // last slice is staged masked and repaired (StreamOperand::mask / fix), exact descriptor ranges
    static constexpr int ABL = ABL_;   // measurement-only: 1 = no refills (LDS + MFMA only), 2 = no MFMA (memory path only),
                                       // 3 = full kernel + wait-time accounting (slots 8-10 of the timing buffer),
                                       // 4 = full kernel (correct results), data movers at s_setprio 3
                                       // 5 = full kernel (correct results), operands streamed with the nontemporal policy
    static constexpr int TM = BM / 32, TN = BN / 32;    // 16 x 16 fragments per wave (2 x 2 waves)
};

#define CTAMD_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// Workgroup = 8 waves: waves 0-3 multiply (one per SIMD, each owns a quarter of the output tile),
// waves 4-7 only move data (one per SIMD, each owns a quarter of every tile's LDS-DMA pieces).  An
// LDS-DMA instruction holds its wave's issue port for 60-100+ cycles; on a wave of its own that time
// overlaps the multiplying wave's MFMAs instead of draining its one-deep MFMA queue.
// Both roles execute exactly nTiles workgroup barriers:
//   barrier #0      : tile 0 has landed
//   barrier #(t+1)  : tile t+1 has landed (loaders waited for their pieces) and every multiplying wave
//                     has finished reading tile t (its fragments are in registers) -> slot t % S is free
template <class Cfg>
__global__ void __launch_bounds__(512, 2) gett_f32_stream_kernel(const GettParams p) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = kStreamBK, S = Cfg::S;
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    using OpA = StreamOperand<Cfg::LA, BM>;
    using OpB = StreamOperand<Cfg::LB, BN>;
    constexpr int STAGE = OpA::FLOATS + OpB::FLOATS;
    constexpr int LOADS = OpA::PER_WAVE + OpB::PER_WAVE;   // LDS-DMA instructions per loader wave per tile
    static_assert(S >= 3 && S <= 6 && S * STAGE * 4 <= 160 * 1024, "LDS ring must fit 160 KiB");
    static_assert(LOADS * (S - 1) <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) float lds[S * STAGE];
    prefetch_kernarg<(int)sizeof(GettParams)>();

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;
    const bool loader = wave8 >= 4;
    unsigned long long* tlog = p.timing ? p.timing + (size_t)blockIdx.x * 16 : nullptr;
    // lane id from the exec-mask bit count, not from threadIdx: nothing derived from the launch registers has to stay alive
    // through the main loop for the sake of the epilogue or of a diagnostic stamp
    auto lane_now = []() -> int {
#if defined(__HIP_DEVICE_COMPILE__)
        return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#else
        return 0;
#endif
    };
    auto stamp = [&](int slot) {
        if (tlog != nullptr && wave8 == 0 && lane_now() == 0) tlog[slot] = (slot >= 5 && slot != 7) ? wall_clock64() : __builtin_readcyclecounter();
    };
    stamp(0);
    stamp(5);

    uint32_t id = xcd_remap(blockIdx.x, p.nBlocks);
    const uint32_t mt = id % p.tilesM; id /= p.tilesM;
    const uint32_t nt = id % p.tilesN; id /= p.tilesN;
    const uint32_t slice = id % p.splitK;
    const uint32_t l = id / p.splitK;
    const uint32_t m0 = mt * BM, n0 = nt * BN;
    // K range of this workgroup (fast-K: every tile is full; any tile count >= 1).  Uniform split: slice *
    // kPerSlice.  Balanced split (p.xcdTiles != 0, one output tile, splitK % 8 == 0): the slices of XCD x —
    // ids [x q, (x+1) q), q = splitK / 8, after the XCD remap — hold xcdTiles.byte[x] tiles each, so that a
    // die whose sustained clock is lower gets proportionally less of K (host: calibrate_xcd_split).
    uint32_t kBegin = slice * p.kPerSlice;
    int nTiles;
    if (p.xcdTiles != 0) {
        const uint32_t q = p.splitK >> 3;
        const uint32_t x = slice / q, j = slice - x * q;
        uint32_t first = 0, mine = 0;
#pragma unroll
        for (uint32_t y = 0; y < 8; ++y) {
            const uint32_t ny = (uint32_t)(p.xcdTiles >> (8 * y)) & 0xffu;
            first += (y < x) ? ny * q : 0u;
            mine = (y == x) ? ny : mine;
        }
        kBegin = (first + j * mine) * BK;
        nTiles = (int)mine;
    } else {
        uint32_t kEnd = kBegin + p.kPerSlice;
        if (kEnd > p.gK.total) kEnd = p.gK.total;
        nTiles = (int)((kEnd - kBegin + (Cfg::RAG ? (uint32_t)BK - 1u : 0u)) / BK);
    }

    if (loader) {
        // =========================== data movers ======================================================
        const float* baseA = static_cast<const float*>(p.A) + group_offset<0>(p.gL, l);
        const float* baseB = static_cast<const float*>(p.B) + group_offset<1>(p.gL, l);
        // RAG: descriptors that end with the tensor (spans below 2^31 bytes: rank_contraction_choices)
        const BufRsrc A = Cfg::RAG ? make_rsrc(baseA, KOdometer::sgpr((uint32_t)(p.endA - (unsigned long long)(uintptr_t)baseA))) : make_rsrc(baseA);
        const BufRsrc B = Cfg::RAG ? make_rsrc(baseB, KOdometer::sgpr((uint32_t)(p.endB - (unsigned long long)(uintptr_t)baseB))) : make_rsrc(baseB);
        if constexpr (Cfg::ABL == 4) __builtin_amdgcn_s_setprio(3);   // experiment: data movers outrank the multipliers
        OpA oa;
        OpB ob;
        oa.template init<0>(p.gM, p.gK, m0, wave, lane);
        ob.template init<1>(p.gN, p.gK, n0, wave, lane);
        KOdometer odo;
        odo.template init<Cfg::RAG>(p.gK, kBegin);
        // RAG: the tile (index among this workgroup's) that is staged masked — the last K-tile of the last slice — and the k it holds
        const uint32_t kTilesAll = (p.gK.total + (uint32_t)BK - 1u) / (uint32_t)BK;
        const int maskAt = (Cfg::RAG && kBegin / (uint32_t)BK + (uint32_t)nTiles == kTilesAll) ? nTiles - 1 : 0x7fffffff;
        const uint32_t kValid = KOdometer::sgpr((p.gK.total % (uint32_t)BK) != 0u ? p.gK.total % (uint32_t)BK : (uint32_t)BK);
        int issued = 0;
        auto issue = [&](int slot) {
            float* stage = lds + slot * STAGE;
            constexpr int AUX = (Cfg::ABL == 5) ? 2 : 0;       // 5 = correct results, nontemporal operand stream
            if constexpr (Cfg::RAG) {
                if (issued == maskAt) { oa.mask(wave, lane, kValid); ob.mask(wave, lane, kValid); }
                ++issued;
            }
            oa.template issue<AUX>(A, odo.offA, stage, wave);
            ob.template issue<AUX>(B, odo.offB, stage + OpA::FLOATS, wave);
            odo.advance(p.gK);
        };
        // the masked tile has landed (this wave's pieces: it waited for them): zero the tail of partial k-units, in front of the barrier
        auto fix_last = [&]() {
            if constexpr (Cfg::RAG) {
                if (maskAt != 0x7fffffff && (kValid & 3u) != 0u) {
                    float* stage = lds + (maskAt % S) * STAGE;
                    oa.fix(stage, wave, lane, kValid);
                    ob.fix(stage + OpA::FLOATS, wave, lane, kValid);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        };
        if (tlog != nullptr && tid == 256) tlog[7] = __builtin_readcyclecounter();   // setup done, first issue
        // Progressive start: the multiplying waves are released as soon as tile 0 has landed, while the
        // rest of the ring is still being requested (an LDS-DMA issue that misses the TLB takes hundreds
        // of cycles, so S tiles of issue time in front of barrier #0 would be S times the start-up cost).
        issue(0);
        if (nTiles > 1) {
            issue(1);
            CTAMD_WAIT_VMCNT(LOADS);
        } else {
            CTAMD_WAIT_VMCNT(0);
            fix_last();                                    // ONE tile: it is the masked one
        }
        __builtin_amdgcn_s_barrier();                      // #0
#pragma unroll
        for (int T = 2; T < S; ++T)
            if (T < nTiles) issue(T);
        int slot = 0, t = 0;
        unsigned long long waitV = 0, waitB = 0;
        for (; t + S < nTiles; ++t) {                      // outstanding: tiles t+1 .. t+S-1
            unsigned long long c0 = 0, c1 = 0, c2 = 0;
            if constexpr (Cfg::ABL == 3) c0 = __builtin_readcyclecounter();
            if constexpr (Cfg::ABL == 1) CTAMD_WAIT_VMCNT(0); else CTAMD_WAIT_VMCNT(LOADS * (S - 2));
            if constexpr (Cfg::ABL == 3) c1 = __builtin_readcyclecounter();
            __builtin_amdgcn_s_barrier();                  // #(t+1): slot t % S is free
            if constexpr (Cfg::ABL == 3) { c2 = __builtin_readcyclecounter(); waitV += c1 - c0; waitB += c2 - c1; }
            if constexpr (Cfg::ABL != 1) issue(slot);
            slot = (slot + 1 == S) ? 0 : slot + 1;
        }
        if constexpr (Cfg::ABL == 3) {
            if (tlog != nullptr && tid == 256) { tlog[9] = waitV; tlog[10] = waitB; }
        }
        // every tile is on its way: one barrier per remaining tile, waiting for exactly the tiles behind it
        for (; t + 1 < nTiles; ++t) {
            const int behind = nTiles - t - 2;             // tiles issued after tile t+1: 0 .. S-2
            if (behind <= 0) { CTAMD_WAIT_VMCNT(0); fix_last(); }   // tile t + 1 is the last one
            else if (behind == 1) CTAMD_WAIT_VMCNT(LOADS);
            else if (behind == 2) CTAMD_WAIT_VMCNT(LOADS * 2);
            else if (behind == 3) CTAMD_WAIT_VMCNT((S > 4 ? LOADS * 3 : 0));
            else CTAMD_WAIT_VMCNT((S > 5 ? LOADS * 4 : 0));
            __builtin_amdgcn_s_barrier();                  // #(t+1)
        }
        CTAMD_WAIT_VMCNT(0);
        return;
    }

    // =============================== multipliers ======================================================
    __builtin_amdgcn_s_setprio(2);
    const int wm = wave & 1, wn = wave >> 1;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int baseA[TM], baseB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) baseA[i] = OpA::frag_base(wm * (BM / 2) + 16 * i, lane);
#pragma unroll
    for (int j = 0; j < TN; ++j) baseB[j] = OpB::frag_base(wn * (BN / 2) + 16 * j, lane);

    f32x4 a0[TM], b0[TN], a1[TM], b1[TN];   // fragments of the even / odd 16-step
    auto load0 = [&](const float* st) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = OpA::template fragment<0>(st, baseA[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = OpB::template fragment<0>(st + OpA::FLOATS, baseB[j]);
    };
    auto load1 = [&](const float* st) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a1[i] = OpA::template fragment<1>(st, baseA[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b1[j] = OpB::template fragment<1>(st + OpA::FLOATS, baseB[j]);
    };
    // MFMAs [FIRST, LAST) of one 16-step, numbered kk-major so that consecutive MFMAs never share an
    // accumulator (dependent latency 40 cycles > issue interval 32)
#define CTAMD_MFMA_RANGE(FA, FB, FIRST, LAST)                                                          \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                     \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
        const int idx = (kk * TM + i) * TN + j;                                                        \
        if (Cfg::ABL != 2 && idx >= (FIRST) && idx < (LAST))                                           \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(FA[i][kk], FB[j][kk], acc[i][j], 0, 0, 0); \
    }
    constexpr int NMFMA = 4 * TM * TN;          // MFMAs per 16-step
    constexpr int SPLIT = NMFMA / 3;            // MFMAs of the odd step issued before the barrier
    // scheduling hint: n x (1 MFMA, 1 LDS read)
#define CTAMD_INTERLEAVE_DS(n)                                         \
    _Pragma("unroll") for (int z = 0; z < (n); ++z) {                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             \
    }

    __builtin_amdgcn_s_barrier();                          // #0: tile 0 has landed
    __builtin_amdgcn_sched_barrier(0);
    load0(lds);
    stamp(1);
    unsigned long long waitC = 0;   // ABL == 3: cycles this wave spent in the per-tile barrier

    // One K-tile in ring slot U (compile-time, so every LDS address is base + immediate).  The fragments
    // of the odd 16-step are fetched under the MFMAs of the even step; the barrier sits a third into the
    // odd step and the first fragments of the next tile are fetched under the remaining two thirds.
#define CTAMD_TILE_BODY(U, LASTTILE) CTAMD_TILE_BODY_AT(lds + (U) * STAGE, lds + (((U) + 1) % S) * STAGE, LASTTILE)
#define CTAMD_TILE_BODY_AT(CUR, NXT, LASTTILE) CTAMD_TILE_BODY_AT2(CUR, NXT, LASTTILE, load0, load1)
#define CTAMD_TILE_BODY_AT2(CUR, NXT, LASTTILE, load0, load1)                                              \
    {                                                                                                      \
        const float* cur = (CUR);                                                                          \
        const float* nxt = (NXT);                                                                          \
        load1(cur);                                                                                        \
        CTAMD_MFMA_RANGE(a0, b0, 0, NMFMA)                                                                 \
        CTAMD_INTERLEAVE_DS(TM + 4 * TN)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        CTAMD_MFMA_RANGE(a1, b1, 0, SPLIT)                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if constexpr (!(LASTTILE)) {                                                                       \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's reads of the tile are back */ \
            unsigned long long cb0 = 0;                                                                    \
            if constexpr (Cfg::ABL == 3) cb0 = __builtin_readcyclecounter();                               \
            __builtin_amdgcn_s_barrier();                                                                  \
            if constexpr (Cfg::ABL == 3) waitC += __builtin_readcyclecounter() - cb0;                      \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            load0(nxt);                                                                                    \
            CTAMD_MFMA_RANGE(a1, b1, SPLIT, NMFMA)                                                         \
            CTAMD_INTERLEAVE_DS(TM + 4 * TN)                                                               \
        } else {                                                                                           \
            CTAMD_MFMA_RANGE(a1, b1, SPLIT, NMFMA)                                                         \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
#define CTAMD_BODY_MID(U) CTAMD_TILE_BODY(U, false)
    // compile-time unrolling over the ring slots
#define CTAMD_FOR_SLOTS(M)                                                     \
    { M(0) M(1) M(2)                                                           \
      if constexpr (S > 3) { M(3) } if constexpr (S > 4) { M(4) }              \
      if constexpr (S > 5) { M(5) } }
    // whole ring turns whose S tiles all have a successor, then the last 1 .. S tiles (slots 0 .. r-1)
    int t = 0;
    for (; t + S < nTiles; t += S) CTAMD_FOR_SLOTS(CTAMD_BODY_MID)
    stamp(2);
    const int r = nTiles - t;
#define CTAMD_BODY_END(U) CTAMD_TILE_BODY(U, (U) == S - 1)
    if (r == S) {          // the common case (whole ring turns): straight-line code, no per-tile branch
        CTAMD_FOR_SLOTS(CTAMD_BODY_END)
    } else {
        // 1 .. S - 1 tiles left (slots 0 .. r - 1): ONE rolled copy of the tile body with run-time slot addresses.  (Unrolled
        // per slot with a branch on r in front of every copy, this tail alone spilled 90-180 VGPRs to scratch memory in the
        // 128 x 128 instantiations — and a kernel that spills is one the next unrelated edit can break.)
        // The fragment bases go through an opaque copy per use, so that derived addresses (base ^ 16, base + slot) are formed
        // where they are needed instead of being carried through the loop in registers it does not have.
        auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
        auto load0t = [&](const float* st) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a0[i] = OpA::template fragment<0>(st, opaque(baseA[i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) b0[j] = OpB::template fragment<0>(st + OpA::FLOATS, opaque(baseB[j]));
        };
        auto load1t = [&](const float* st) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a1[i] = OpA::template fragment<1>(st, opaque(baseA[i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) b1[j] = OpB::template fragment<1>(st + OpA::FLOATS, opaque(baseB[j]));
        };
        int u = 0;
#pragma unroll 1
        for (; u + 1 < r; ++u) CTAMD_TILE_BODY_AT2(lds + u * STAGE, lds + (u + 1) * STAGE, false, load0t, load1t)
        CTAMD_TILE_BODY_AT2(lds + u * STAGE, lds, true, load0t, load1t)     // u == r - 1
    }
    stamp(3);
    if constexpr (Cfg::ABL == 3) {
        if (tlog != nullptr && wave8 == 0 && lane_now() == 0) tlog[8] = waitC;
    }

    // ---- epilogue ------------------------------------------------------------------------------------
    // Everything the epilogue derives from the lane number is derived HERE (opaque copy): hoisted to kernel entry it would
    // occupy registers across the main loop, and the 128 x 128 instantiations have none to spare.
    const int laneE = lane_now();
    const int tidE = wave8 * 64 + laneE;           // multiplying waves: wave8 = 0..3
    const uint32_t Mtot = p.gM.total, Ntot = p.gN.total;
    if (p.partial != nullptr) {
        // accumulator-order partials: [slice][l][nt][mt][wave][i][j][lane] x 16 B
        const size_t tileIdx = (((size_t)slice * p.gL.total + l) * p.tilesN + nt) * p.tilesM + mt;
        // write-through (sc1) stores: the lines leave the XCD's L2 while other workgroups still multiply,
        // so the kernel boundary in front of the fold kernel has no dirty partials left to flush
        f32x4* P = reinterpret_cast<f32x4*>(p.partial) + (tileIdx * 4 + wave) * (size_t)(TM * TN * 64);
        const BufRsrc rP = make_rsrc(reinterpret_cast<const float*>(P));
        if (__builtin_expect(p.partialPolicy != 0, 0)) {      // measurement switch (wave-uniform): plain or nontemporal stores
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (p.partialPolicy == 1) store_policy_16<0>(acc[i][j], rP, (uint32_t)(((i * TN + j) * 64 + laneE) * 16));
                    else store_policy_16<2>(acc[i][j], rP, (uint32_t)(((i * TN + j) * 64 + laneE) * 16));
                }
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) store_wt_16(acc[i][j], rP, (uint32_t)(((i * TN + j) * 64 + laneE) * 16));
        }
        stamp(4);
        if (p.sync == nullptr) {   // the fold runs as its own kernel
            stamp(6);
            return;
        }
        // ---- in-launch fold (planner: one workgroup per CU, all co-resident; flat M / N / no batch) ------
        // Publish: the partial stores above are write-through; every storing wave drains them, the four
        // multiplying waves meet (the data movers have ended), one lane bumps the arrival counter with an
        // agent-scope atomic.  Consume: that lane polls the counter relaxed, ONE agent-scope acquire drops
        // this CU's stale L1 lines, the workgroup meets again and then reads its share of every slice with
        // plain loads (cdna_hip_programming.md Guideline 16, recipe R1).  The wait is bounded: a grid that is
        // not co-resident yields wrong numbers, never a hang.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        uint32_t* cnt = p.sync;
        if (tidE == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.nBlocks && ++spins < (1u << 22))
                __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            // the last workgroup through re-arms the counters for the next launch that draws this slot
            if (__hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.nBlocks - 1) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __builtin_amdgcn_s_barrier();
        stamp(11);
        // share: 128-byte lines (8 accumulator quads) of the tile image, line i -> workgroup i mod nBlocks
        const uint32_t quadsTotal = p.tilesM * p.tilesN * 4u * TM * TN * 64u;
        const uint32_t lines = quadsTotal / 8u;
        const int q = tidE & 7, g = tidE >> 3;                // 8 quads x 32 slice groups (256 multiplying threads)
        f32x4* red = reinterpret_cast<f32x4*>(lds);         // the ring is idle now
        for (uint32_t line = blockIdx.x; line < lines; line += p.nBlocks) {
            const uint32_t e = line * 8u + q;
            const f32x4* src = reinterpret_cast<const f32x4*>(p.partial) + e;
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            for (uint32_t s0 = g; s0 < p.splitK; s0 += 256) {
                f32x4 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t sl = s0 + 32u * u;
                    x[u] = (sl < p.splitK) ? __builtin_nontemporal_load(src + (size_t)sl * quadsTotal) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                sum += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
            }
#pragma unroll
            for (int m = 8; m < 64; m <<= 1)
#pragma unroll
                for (int c = 0; c < 4; ++c) sum[c] += __shfl_xor(sum[c], m, 64);
            if (laneE < 8) red[wave * 8 + q] = sum;
            __builtin_amdgcn_s_barrier();
            if (tidE < 8) {
                sum = (red[q] + red[8 + q]) + (red[16 + q] + red[24 + q]);
                uint32_t rem = e;
                const uint32_t ln = rem % 64; rem /= 64;
                const uint32_t fj = rem % TN; rem /= TN;
                const uint32_t fi = rem % TM; rem /= TM;
                const uint32_t w = rem % 4; rem /= 4;
                const uint32_t tmt = rem % p.tilesM;
                const uint32_t tnt = rem / p.tilesM;
                const uint32_t n = tnt * BN + (w >> 1) * (BN / 2) + 16 * fj + (ln & 15);
                if (n < Ntot) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t m = tmt * BM + (w & 1) * (BM / 2) + 16 * fi + 4 * (ln >> 4) + r;
                        if (m >= Mtot) continue;
                        float val = p.alpha * sum[r];
                        if (p.beta != 0.f)
                            val += p.beta * static_cast<const float*>(p.C)[(int64_t)m * p.cStrideM[0] + (int64_t)n * p.cStrideN[0]];
                        static_cast<float*>(p.D)[(int64_t)m * p.gM.stride[1][0] + (int64_t)n * p.gN.stride[1][0]] = val;
                    }
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        stamp(6);
        return;
    }
    // Outputs with 16-byte lanes along N leave as whole rows through a per-wave LDS image (gett_common.h, round 6).  The
    // image lives in the ring slot BEHIND the last tile's: every multiplying wave passed the last tile's barrier, so every older slot has
    // been read for the last time (S >= 3), while the last tile's own slot may still be feeding a slower wave's fragments.
    {
        constexpr int IMG = gett_f32_image_floats<TN>();
        static_assert(4 * IMG <= STAGE, "four per-wave row images fit one ring slot");
        const GettArgPtr q = gett_arg_ptr();
        const float* Cl;
        float* Dl;
        if (gett_f32_rows_ok(q, l, Cl, Dl)) {
            gett_store_tile_f32_rows<TM, TN>(q, Cl, Dl, acc, m0 + wm * (BM / 2), n0 + wn * (BN / 2), laneE, lds + (size_t)(nTiles % S) * STAGE + wave * IMG);
            stamp(4);
            stamp(6);
            return;
        }
    }
    const float* C = static_cast<const float*>(p.C);
    float*       D = static_cast<float*>(p.D);
    {
        int64_t oD, oC;
        group_offset2<2>(p.gL, p.cStrideL, l, oD, oC);
        D += oD;
        C += oC;
    }
    int64_t offDn[TN], offCn[TN];
    bool    okN[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const uint32_t n = n0 + wn * (BN / 2) + 16 * j + (laneE & 15);
        okN[j] = n < Ntot;
        offDn[j] = 0;
        offCn[j] = 0;
        if (okN[j]) group_offset2<1>(p.gN, p.cStrideN, n, offDn[j], offCn[j]);
    }
    const float alpha = p.alpha, beta = p.beta;
    // A lane's four accumulator registers of a fragment are four consecutive m at one n.  When the fastest M mode of D is
    // contiguous, a multiple of 4 long and everything else keeps 16-byte alignment (wave-uniform test), they leave as ONE
    // 16-byte store (64-byte row pieces per instruction) instead of four 4-byte stores — 4-byte pieces cost ~6x the time per
    // byte (tools/f32_ksweep.py: fixed cost per workgroup).  Nontemporal: the result is not read again, the operand panels
    // stay in L2.
    bool vecD = p.gM.stride[1][0] == 1 && (p.gM.div[0].d & 3u) == 0u && (reinterpret_cast<uintptr_t>(D) & 15u) == 0u;
    bool vecC = vecD && p.cStrideM[0] == 1 && (reinterpret_cast<uintptr_t>(C) & 15u) == 0u;
#pragma unroll
    for (int q = 0; q < kMaxGroupModes; ++q) {
        vecD = vecD && (q == 0 || (p.gM.stride[1][q] & 3) == 0) && (p.gN.stride[1][q] & 3) == 0;
        vecC = vecC && (q == 0 || (p.cStrideM[q] & 3) == 0) && (p.cStrideN[q] & 3) == 0;
    }
    if (vecD) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint32_t m = m0 + wm * (BM / 2) + 16 * i + 4 * (laneE >> 4);
            if (m >= Mtot) continue;               // Mtot is a multiple of 4 here: the four rows are all in or all out
            int64_t offDm, offCm;
            group_offset2<1>(p.gM, p.cStrideM, m, offDm, offCm);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (!okN[j]) continue;
                f32x4 val = {alpha * acc[i][j][0], alpha * acc[i][j][1], alpha * acc[i][j][2], alpha * acc[i][j][3]};
                if (beta != 0.f) {
                    const float* c = C + offCm + offCn[j];
                    if (vecC) {
                        const f32x4 cv = *reinterpret_cast<const f32x4*>(c);
                        val[0] += beta * cv[0]; val[1] += beta * cv[1]; val[2] += beta * cv[2]; val[3] += beta * cv[3];
                    } else {
                        int64_t dD, dC1, dC2, dC3;
                        group_offset2<1>(p.gM, p.cStrideM, m + 1, dD, dC1);
                        group_offset2<1>(p.gM, p.cStrideM, m + 2, dD, dC2);
                        group_offset2<1>(p.gM, p.cStrideM, m + 3, dD, dC3);
                        val[0] += beta * c[0];
                        val[1] += beta * C[dC1 + offCn[j]];
                        val[2] += beta * C[dC2 + offCn[j]];
                        val[3] += beta * C[dC3 + offCn[j]];
                    }
                }
                __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(D + offDm + offDn[j]));
            }
        }
        stamp(4);
        stamp(6);
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t m = m0 + wm * (BM / 2) + 16 * i + 4 * (laneE >> 4) + r;
            if (m >= Mtot) continue;
            int64_t offDm, offCm;
            group_offset2<1>(p.gM, p.cStrideM, m, offDm, offCm);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (!okN[j]) continue;
                float val = alpha * acc[i][j][r];
                if (beta != 0.f) val += beta * C[offCm + offCn[j]];
                D[offDm + offDn[j]] = val;
            }
        }
    stamp(4);
    stamp(6);
}

// ---------------------------------------------------------------------------------------------
// Split-K fold for accumulator-order partials: D = alpha * sum_s partial[s] + beta * C.
// One lane owns one 16-byte accumulator register quad (4 consecutive m at one n) of one output tile.
// A workgroup covers 8 quads (one 128-byte line per slice) x 32 slice groups; every lane has all its
// loads (splitK / 32 of them, 8 for a 256-way split) in flight before the first add, the slice groups
// of a wave meet through lane shuffles and the four waves through LDS.  Latency-bound, not
// bandwidth-bound (9.4 MB for the headline einsum): many small workgroups, one memory round trip.
// The sum order is fixed by the layout, so results are reproducible run to run.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) splitk_reduce_frag_kernel(const SplitKReduceParams p) {
    __shared__ f32x4 red[4][8];
    const int q = threadIdx.x & 7, g = threadIdx.x >> 3;     // g = 0..31, 8 groups per wave
    const uint32_t quadsPerTile = 4u * p.fragTM * p.fragTN * 64u;
    const size_t   tilesTotal = (size_t)p.gL.total * p.tilesN * p.tilesM;
    const size_t   quadsTotal = tilesTotal * quadsPerTile;
    const size_t   e = (size_t)blockIdx.x * 8 + q;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (e < quadsTotal) {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.partial) + e;
        for (uint32_t s0 = g; s0 < p.splitK; s0 += 256) {
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t sl = s0 + 32u * u;
                x[u] = (sl < p.splitK) ? __builtin_nontemporal_load(src + (size_t)sl * quadsTotal) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            sum += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
        }
    }
    // the 8 slice groups of a wave: lanes that differ in bits 3..5
#pragma unroll
    for (int m = 8; m < 64; m <<= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) sum[c] += __shfl_xor(sum[c], m, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 8) red[wave][q] = sum;
    __syncthreads();
    if (threadIdx.x >= 8 || e >= quadsTotal) return;
    sum = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
    // decode e -> (l, nt, mt, wave, i, j, lane)
    size_t rem = e;
    const uint32_t lane = (uint32_t)(rem % 64); rem /= 64;
    const uint32_t j = (uint32_t)(rem % p.fragTN); rem /= p.fragTN;
    const uint32_t i = (uint32_t)(rem % p.fragTM); rem /= p.fragTM;
    const uint32_t w = (uint32_t)(rem % 4); rem /= 4;
    const uint32_t mt = (uint32_t)(rem % p.tilesM); rem /= p.tilesM;
    const uint32_t nt = (uint32_t)(rem % p.tilesN); rem /= p.tilesN;
    const uint32_t l = (uint32_t)rem;
    const uint32_t bm = 32u * p.fragTM, bn = 32u * p.fragTN;
    const uint32_t n = nt * bn + (w >> 1) * (bn / 2) + 16 * j + (lane & 15);
    if (n >= p.gN.total) return;
    int64_t oDl = 0, oCl = 0, oDn, oCn;
    group_offset2<2>(p.gL, p.cStrideL, l, oDl, oCl);
    group_offset2<1>(p.gN, p.cStrideN, n, oDn, oCn);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t m = mt * bm + (w & 1) * (bm / 2) + 16 * i + 4 * (lane >> 4) + r;
        if (m >= p.gM.total) continue;
        int64_t oDm, oCm;
        group_offset2<1>(p.gM, p.cStrideM, m, oDm, oCm);
        float val = p.alpha * sum[r];
        if (p.beta != 0.f) val += p.beta * static_cast<const float*>(p.C)[oCl + oCm + oCn];
        static_cast<float*>(p.D)[oDl + oDm + oDn] = val;
    }
}

// The same fold for the common case of one M mode, one N mode and no batch: every argument fits one
// 96-byte kernel-argument block, fetched in a single round (a dependent round of argument loads costs
// ~0.45 us at the start of a 3-us kernel), and the output address is two multiplies.
struct FoldFlatParams {
    const float* partial;
    const float* C;
    float*       D;
    int64_t      sDm, sDn, sCm, sCn;
    float        alpha, beta;
    uint32_t     splitK, quadsTotal, fragTM, fragTN, tilesM, Mtot, Ntot;
};

// NT threads = 8 quads x NT/8 slice groups; a lane keeps 256 / (NT/8) loads in flight per pass.
template <int NT>
__global__ void __launch_bounds__(NT) splitk_reduce_frag_flat_kernel(const FoldFlatParams p) {
    constexpr int G = NT / 8, U = 256 / G, W = NT / 64;
    __shared__ f32x4 red[W][8];
    const int q = threadIdx.x & 7, g = threadIdx.x >> 3;
    const uint32_t e = blockIdx.x * 8 + q;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (e < p.quadsTotal) {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.partial) + e;
        for (uint32_t s0 = g; s0 < p.splitK; s0 += 256) {
            f32x4 x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t sl = s0 + (uint32_t)G * u;
                x[u] = (sl < p.splitK) ? __builtin_nontemporal_load(src + (size_t)sl * p.quadsTotal) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int st = 1; st < U; st <<= 1)
#pragma unroll
                for (int u = 0; u + st < U; u += 2 * st) x[u] += x[u + st];
            sum += x[0];
        }
    }
#pragma unroll
    for (int m = 8; m < 64; m <<= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) sum[c] += __shfl_xor(sum[c], m, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 8) red[wave][q] = sum;
    __syncthreads();
    if (threadIdx.x >= 8 || e >= p.quadsTotal) return;
    sum = red[0][q];
#pragma unroll
    for (int w = 1; w < W; ++w) sum += red[w][q];
    uint32_t rem = e;
    const uint32_t lane = rem % 64; rem /= 64;
    const uint32_t j = rem % p.fragTN; rem /= p.fragTN;
    const uint32_t i = rem % p.fragTM; rem /= p.fragTM;
    const uint32_t w = rem % 4; rem /= 4;
    const uint32_t mt = rem % p.tilesM;
    const uint32_t nt = rem / p.tilesM;
    const uint32_t bm = 32u * p.fragTM, bn = 32u * p.fragTN;
    const uint32_t n = nt * bn + (w >> 1) * (bn / 2) + 16 * j + (lane & 15);
    if (n >= p.Ntot) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t m = mt * bm + (w & 1) * (bm / 2) + 16 * i + 4 * (lane >> 4) + r;
        if (m >= p.Mtot) continue;
        float val = p.alpha * sum[r];
        if (p.beta != 0.f) val += p.beta * p.C[(int64_t)m * p.sCm + (int64_t)n * p.sCn];
        p.D[(int64_t)m * p.sDm + (int64_t)n * p.sDn] = val;
    }
}

hipError_t launch_splitk_reduce_frag(const SplitKReduceParams& p, hipStream_t stream) {
    const size_t quads = (size_t)p.gL.total * p.tilesN * p.tilesM * 4u * p.fragTM * p.fragTN * 64u;
    const size_t blocks = (quads + 7) / 8;
    if (blocks == 0) return hipSuccess;
    if (p.gL.total == 1 && p.gM.n <= 1 && p.gN.n <= 1 && quads < (1ull << 31)) {
        FoldFlatParams f;
        f.partial = p.partial;
        f.C = static_cast<const float*>(p.C);
        f.D = static_cast<float*>(p.D);
        f.sDm = p.gM.stride[1][0]; f.sDn = p.gN.stride[1][0];
        f.sCm = p.cStrideM[0];     f.sCn = p.cStrideN[0];
        f.alpha = p.alpha; f.beta = p.beta;
        f.splitK = p.splitK; f.quadsTotal = (uint32_t)quads;
        f.fragTM = p.fragTM; f.fragTN = p.fragTN; f.tilesM = p.tilesM;
        f.Mtot = p.gM.total; f.Ntot = p.gN.total;
        // 128 / 512 / 1024 threads per workgroup measured the same step time (43.6-44.0 us): the kernel is one memory
        // round trip plus launch, not throughput
        hipLaunchKernelGGL(splitk_reduce_frag_flat_kernel<256>, dim3((unsigned)blocks), dim3(256), 0, stream, f);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(splitk_reduce_frag_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Kernel table
// ---------------------------------------------------------------------------------------------
template <class Cfg>
static hipError_t launch_stream(const GettParams& p, hipStream_t stream) {
    if constexpr (Cfg::ABL == 0) {
        // ragged K (one contracted mode) or operands without 16-byte lanes (rank_contraction_choices): the RAG twin of this tile on the
        // 4-deep ring — masked + repaired last K-tile, descriptors that end with the tensor
        if (p.gK.total % (uint32_t)kStreamBK != 0u || (p.ragged & 1u) != 0u) {
            using R = StreamCfg<Cfg::BM, Cfg::BN, Cfg::LA, Cfg::LB, 4, 0, true>;
            hipLaunchKernelGGL(gett_f32_stream_kernel<R>, dim3(p.nBlocks), dim3(512), 0, stream, p);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(gett_f32_stream_kernel<Cfg>, dim3(p.nBlocks), dim3(512), 0, stream, p);
    return hipGetLastError();
}

// XS(bm, bn, layA, layB, stages)
#define CTAMD_STREAM_KERNELS(XS)        \
    XS(96, 96, LAY_K, LAY_F, 4)         \
    XS(96, 96, LAY_K, LAY_F, 6)         \
    XS(96, 96, LAY_K, LAY_F, 3)         \
    XS(96, 96, LAY_F, LAY_F, 4)         \
    XS(96, 96, LAY_F, LAY_K, 4)         \
    XS(96, 96, LAY_K, LAY_K, 4)         \
    XS(128, 128, LAY_F, LAY_F, 4)       \
    XS(128, 128, LAY_K, LAY_F, 4)       \
    XS(128, 128, LAY_F, LAY_K, 4)       \
    XS(128, 128, LAY_K, LAY_K, 4)       \
    XS(64, 64, LAY_F, LAY_F, 4)         \
    XS(64, 64, LAY_K, LAY_F, 4)         \
    XS(64, 64, LAY_F, LAY_K, 4)         \
    XS(64, 64, LAY_K, LAY_K, 4)         \
    XS(96, 96, LAY_F, LAY_F, 3)         \
    XS(96, 96, LAY_F, LAY_K, 3)         \
    XS(96, 96, LAY_K, LAY_K, 3)         \
    XS(128, 128, LAY_F, LAY_F, 3)       \
    XS(128, 128, LAY_K, LAY_F, 3)

#define CTAMD_STREAM_ENTRY(bm, bn, la, lb, s) \
    {bm, bn, kStreamBK, 2, 2, 1, la, lb, 512, s, 1, 0, &launch_stream<StreamCfg<bm, bn, la, lb, s>>, 1},

#define CTAMD_STREAM_ABL(bm, bn, la, lb, s, abl) \
    {bm, bn, kStreamBK, 2, 2, 1, la, lb, 512, s, 1, abl, &launch_stream<StreamCfg<bm, bn, la, lb, s, abl>>, 1},

#define CTAMD_STREAM_NT(bm, bn, la, lb, s) \
    {bm, bn, kStreamBK, 2, 2, 1, la, lb, 512, s, 1, 0, &launch_stream<StreamCfg<bm, bn, la, lb, s, 5>>, 1, 1},

static const GettKernelInfo g_stream_table[] = {
    CTAMD_STREAM_KERNELS(CTAMD_STREAM_ENTRY)
    CTAMD_STREAM_ABL(96, 96, LAY_K, LAY_F, 4, 1)
    CTAMD_STREAM_ABL(96, 96, LAY_K, LAY_F, 4, 2)
    CTAMD_STREAM_ABL(96, 96, LAY_K, LAY_F, 4, 3)
    CTAMD_STREAM_ABL(96, 96, LAY_K, LAY_F, 4, 4)
    // nontemporal operand stream (StreamCfg ABL = 5: correct results): the 3-deep 96 x 96 ring for the four layouts.  Measured on the
    // headline shape: operands resident in the Infinity Cache 108 vs 114 TFLOP/s for the default policy (worse), operands from HBM
    // 102-104 vs 99-100 (better) — so the planner ranks these only when the operands cannot be cache-resident (plan_contraction.cpp)
    CTAMD_STREAM_NT(96, 96, LAY_K, LAY_F, 3) CTAMD_STREAM_NT(96, 96, LAY_F, LAY_F, 3)
    CTAMD_STREAM_NT(96, 96, LAY_F, LAY_K, 3) CTAMD_STREAM_NT(96, 96, LAY_K, LAY_K, 3)};

const GettKernelInfo* gett_f32_stream_kernels(int* count) {
    *count = (int)(sizeof(g_stream_table) / sizeof(g_stream_table[0]));
    return g_stream_table;
}

}  // namespace ctamd
