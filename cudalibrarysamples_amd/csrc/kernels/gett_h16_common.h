// gett_h16_common.h — shared pieces of the 16-bit GETT kernels (gett_h16.hip, gett_h16v.hip): types, the LDS-DMA / fragment-read
// primitives, the operand staging tables (HOperand), the K odometer and the epilogue.  See gett_h16.hip for the LDS image.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "params.h"
#include "launch.h"
#include "gett_common.h"

namespace ctamd {

constexpr int kHBK   = 64;            // K-tile
constexpr int kHTile = 256;           // BM = BN
constexpr int kHalfBytes = 16384;     // one half-tile: 128 rows x 64 k x 2 B

typedef short    s16x4 __attribute__((ext_vector_type(4)));
typedef short    s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));

// Raw buffer descriptor (stride 0, num_records 2^32 - 1, gfx9 raw-buffer format word), built from
// readfirstlane'd words so that the compiler knows it is wave-uniform: an inline-asm "s" operand that is
// not provably uniform is silently given VGPRs.
typedef int HRsrc __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t h_uniform64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// descriptor whose base is the (wave-uniform) byte address `addr`
__device__ __forceinline__ HRsrc h_make_rsrc(uint64_t addr) {
    HRsrc r;
    r[0] = (int)(uint32_t)addr;
    r[1] = (int)((uint32_t)(addr >> 32) & 0xffffu);
    r[2] = -1;
    r[3] = 0x00020000;
    return r;
}
// Minimum over the 64 lanes of a wave (all active) of byte offsets that lie within 2^31 of each other (the planner admits the
// 16-bit kernels only when ONE tile spans less than 2^31 bytes): the 32-bit differences to the first lane's value are reduced with
// four DPP steps inside each row of 16 lanes and four v_readlane — ~50 cycles.  (As six __shfl_xor rounds on 64-bit values this was
// 24 dependent ds_bpermute per kernel start, ~2.5k cycles of every workgroup's prologue: tools/h16_small_timeline.py.)
__device__ __forceinline__ int64_t h_wave_min(int64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int64_t ref = (int64_t)h_uniform64((uint64_t)v);
    int d = (int)(v - ref);
    int t;
    t = __builtin_amdgcn_update_dpp(d, d, 0xB1, 0xF, 0xF, false);  d = t < d ? t : d;     // quad_perm [1,0,3,2]
    t = __builtin_amdgcn_update_dpp(d, d, 0x4E, 0xF, 0xF, false);  d = t < d ? t : d;     // quad_perm [2,3,0,1]
    t = __builtin_amdgcn_update_dpp(d, d, 0x141, 0xF, 0xF, false); d = t < d ? t : d;     // row_half_mirror
    t = __builtin_amdgcn_update_dpp(d, d, 0x140, 0xF, 0xF, false); d = t < d ? t : d;     // row_mirror: every lane holds its row's minimum
    const int r0 = __builtin_amdgcn_readlane(d, 0), r1 = __builtin_amdgcn_readlane(d, 16);
    const int r2 = __builtin_amdgcn_readlane(d, 32), r3 = __builtin_amdgcn_readlane(d, 48);
    const int m01 = r0 < r1 ? r0 : r1, m23 = r2 < r3 ? r2 : r3;
    return ref + (int64_t)(m01 < m23 ? m01 : m23);
#else
    return v;
#endif
}

// 64 lanes x 16 B -> the 1-KiB LDS piece at byte address ldsByte (wave-uniform).  Hidden from the
// compiler's wait-count bookkeeping on purpose: completion is counted by hand (CTAMD_H_VMCNT).
// s_nop 4: the SGPR operands may come straight from a v_readfirstlane (VALU-write -> VMEM-read hazard).
template <bool PAD = true>
__device__ __forceinline__ void h_dma16(HRsrc rsrc, uint32_t laneBytes, uint32_t ldsByte) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PAD)
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     :: "s"(ldsByte), "v"(laneBytes), "s"(rsrc) : "memory");
    else   // main loop: every SGPR operand was produced by the scalar ALU, or long ago
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                     :: "s"(ldsByte), "v"(laneBytes), "s"(rsrc) : "memory");
#else
    (void)rsrc; (void)laneBytes; (void)ldsByte;
#endif
}

#define CTAMD_H_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define CTAMD_H_LGKM0()  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <bool BF>
__device__ __forceinline__ f32x16 h_mfma(s16x8 a, s16x8 b, f32x16 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else              return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
    (void)a; (void)b; return c;
#endif
}

__device__ __forceinline__ float h_to_float(uint16_t v, bool bf) {
    if (bf) return __uint_as_float((uint32_t)v << 16);
    return (float)__builtin_bit_cast(_Float16, v);
}
__device__ __forceinline__ uint16_t h_from_float(float f, bool bf) {
    if (bf) {   // round to nearest even; NaN stays NaN
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    }
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of the 16-bit kernels: D = alpha * acc + beta * C, one rounding to the 16-bit type.
// An MFMA accumulator fragment holds one output column per lane (32 consecutive n in lanes 0-31, 16 rows in the 16
// registers).  Storing it from the registers means 16 two-byte stores per lane and fragment (two-byte stores cost ~12x the
// time per byte of 16-byte ones), and — worse — fully unrolled address arithmetic and conversions for every one of them:
// the first form of this epilogue was 100+ KB of straight-line code that every workgroup streamed through once, and the
// instruction fetch alone cost 30 us of a workgroup's 150 us on 8192^3 (K-sweep: 37 us fixed cost per workgroup, 7 us
// without the epilogue).  Now a wave parks four fragments at a time in 16 KiB of LDS of its own (the operand ring is dead by
// then: 64 ds_write_b32 in accumulator order, the only unrolled part) and a ROLLED loop of eight iterations turns them into
// 16-byte row pieces: every lane reads 8 consecutive n of one row, adds beta * C, converts (v_cvt_pk_bf16_f32 / v_cvt_f16_f32)
// and issues one 16-byte global store (rows of 64 contiguous bytes, 4 lanes each).  The vector form needs 8 consecutive n
// contiguous and 16-byte aligned in D (and in C when beta != 0): checked once per workgroup (wave-uniform); otherwise a rolled
// element-wise loop stores from the same LDS image with any strides.
// ---------------------------------------------------------------------------------------------------------------------
template <bool BF>
__device__ __forceinline__ uint16_t h_round16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (BF) return __builtin_bit_cast(uint16_t, (__bf16)f);      // round to nearest even, NaN stays NaN
    else              return __builtin_bit_cast(uint16_t, (_Float16)f);
#else
    (void)f; return 0;
#endif
}

// round16(v + beta * c), c a 16-bit value of the tensor's type, with ONE rounding (round 6).  bf16: the fma is an fp32 operation whatever
// the compiler makes of it (v_fma_f32 / v_pk_fma_f32), then one conversion.  fp16: v_fma_mixlo_f16 by hand — the sum is rounded once,
// straight to 16 bits.  Written as `v += beta * (float)c; (_Float16)v` the compiler picks per element between that instruction and
// v_pk_fma_f32 + v_cvt_pk_f16_f32 (an fp32 sum rounded a second time), depending on how the neighbours vectorise: two kernels of one plan
// (gett_h16w4x_kernel / gett_h16w4p_kernel) then differ in the last bit of a few results per million (profiles/r06w, r06x).
template <bool BF>
__device__ __forceinline__ uint16_t h_round16_with_c(float v, float beta, uint16_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (BF) return h_round16<true>(__builtin_fmaf(beta, __uint_as_float((uint32_t)c << 16), v));
    else {
        uint32_t r;
        const uint32_t cw = c;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(beta), "v"(cw), "v"(v));
        return (uint16_t)r;
    }
#else
    (void)v; (void)beta; (void)c; return 0;
#endif
}
template <bool BF>
__device__ __forceinline__ s16x8 h_round8(const f32x4& v0, const f32x4& v1) {
    return s16x8{(short)h_round16<BF>(v0[0]), (short)h_round16<BF>(v0[1]), (short)h_round16<BF>(v0[2]), (short)h_round16<BF>(v0[3]),
                 (short)h_round16<BF>(v1[0]), (short)h_round16<BF>(v1[1]), (short)h_round16<BF>(v1[2]), (short)h_round16<BF>(v1[3])};
}
template <bool BF>
__device__ __forceinline__ s16x8 h_round8_with_c(const f32x4& v0, const f32x4& v1, float beta, const s16x8& cv) {
    return s16x8{(short)h_round16_with_c<BF>(v0[0], beta, (uint16_t)cv[0]), (short)h_round16_with_c<BF>(v0[1], beta, (uint16_t)cv[1]),
                 (short)h_round16_with_c<BF>(v0[2], beta, (uint16_t)cv[2]), (short)h_round16_with_c<BF>(v0[3], beta, (uint16_t)cv[3]),
                 (short)h_round16_with_c<BF>(v1[0], beta, (uint16_t)cv[4]), (short)h_round16_with_c<BF>(v1[1], beta, (uint16_t)cv[5]),
                 (short)h_round16_with_c<BF>(v1[2], beta, (uint16_t)cv[6]), (short)h_round16_with_c<BF>(v1[3], beta, (uint16_t)cv[7])};
}

// C and D are device memory: the epilogues address them through the GLOBAL address space (global_load / global_store), not through
// generic pointers (flat_*).  A flat operation "may touch LDS" for the compiler's wait-count bookkeeping: inside a kernel that loops
// over tiles (gett_h16p.hip) one pending flat store turned every counted LDS wait of the main loop into lgkmcnt(0).
typedef s16x8 __attribute__((address_space(1))) * HGlbS8;
typedef const s16x8 __attribute__((address_space(1))) * HGlbCS8;
typedef uint16_t __attribute__((address_space(1))) * HGlbU16;
typedef const uint16_t __attribute__((address_space(1))) * HGlbCU16;

struct HEpilogue {
    const uint16_t* C;
    uint16_t*       D;
    float alpha, beta;
    uint32_t Mtot, Ntot;
    bool vecD, vecC;
    bool flat;           // M and N are single modes: offsets are one multiplication, no digit decomposition
    float* scratch;      // this wave's 16 KiB of LDS: four fragments [32 rows][32 n] fp32

    __device__ __forceinline__ void init(const GettParams& p, uint32_t l, char* lds, int wave, int bytesPerWave = 16384) {
        C = static_cast<const uint16_t*>(p.C);
        D = static_cast<uint16_t*>(p.D);
        int64_t oD, oC;
        group_offset2<2>(p.gL, p.cStrideL, l, oD, oC);
        D += oD;
        C += oC;
        alpha = p.alpha; beta = p.beta;
        Mtot = p.gM.total; Ntot = p.gN.total;
        scratch = reinterpret_cast<float*>(lds + wave * bytesPerWave);
        // 16-byte chunks: 8 consecutive n are contiguous and stay inside the fastest N mode — its extent is a multiple of 8, or it is
        // the ONLY N mode (then a row's last chunk may be partial: store16 / load16 below).  No alignment condition (round 6): a
        // 16-byte global access works at any 2-byte address (tools/ubench/ldsdma_unaligned.hip), so a row pitch of 4100 or 4097
        // elements keeps the chunk path instead of 2-byte stores (~12x the time per byte).
        const bool d = p.gN.stride[1][0] == 1 && (p.gN.n <= 1 || p.gN.div[0].d % 8u == 0u);
        vecD = d;
        vecC = d && p.cStrideN[0] == 1;
        flat = p.gM.n <= 1 && p.gN.n <= 1;
    }
    __device__ __forceinline__ void offsets(const GettParams& p, uint32_t m, uint32_t n, int64_t& offD, int64_t& offC) const {
        if (flat) {
            offD = (int64_t)m * p.gM.stride[1][0] + (int64_t)n * p.gN.stride[1][0];
            offC = (int64_t)m * p.cStrideM[0] + (int64_t)n * p.cStrideN[0];
        } else {
            int64_t dm, cm, dn, cn;
            group_offset2<1>(p.gM, p.cStrideM, m, dm, cm);
            group_offset2<1>(p.gN, p.cStrideN, n, dn, cn);
            offD = dm + dn; offC = cm + cn;
        }
    }

    // One 16-byte chunk (columns n .. n + 7 of a row) to D / from C.  A chunk that sticks out of a ragged N mode (n + 8 > Ntot: one N
    // mode, extent % 8 != 0) moves the columns that exist, one element at a time: nothing is read or written past the row.
    __device__ __forceinline__ void store16(uint16_t* dst, const s16x8& v, uint32_t n) const {
        if (n + 8u <= Ntot) { __builtin_nontemporal_store(v, (HGlbS8)(uintptr_t)dst); return; }
#define CTAMD_EP_ST1(E) if (n + (E) < Ntot) *(HGlbU16)(uintptr_t)(dst + (E)) = (uint16_t)v[E];
        CTAMD_EP_ST1(0) CTAMD_EP_ST1(1) CTAMD_EP_ST1(2) CTAMD_EP_ST1(3) CTAMD_EP_ST1(4) CTAMD_EP_ST1(5) CTAMD_EP_ST1(6)
#undef CTAMD_EP_ST1
    }
    __device__ __forceinline__ s16x8 load16(const uint16_t* src, uint32_t n) const {
        if (n + 8u <= Ntot) return *(HGlbCS8)(uintptr_t)src;
        s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#define CTAMD_EP_LD1(E) if (n + (E) < Ntot) v[E] = (short)*(HGlbCU16)(uintptr_t)(src + (E));
        CTAMD_EP_LD1(0) CTAMD_EP_LD1(1) CTAMD_EP_LD1(2) CTAMD_EP_LD1(3) CTAMD_EP_LD1(4) CTAMD_EP_LD1(5) CTAMD_EP_LD1(6)
#undef CTAMD_EP_LD1
        return v;
    }

    // park fragment F (0..3) of the current pass: element (row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31) = acc[r]
    __device__ __forceinline__ void park(int F, const f32x16& acc, int lane) const {
        float* st = scratch + F * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = alpha * acc[r];
    }

    // ---- 64-column form (the default kernel: a wave's two B-side fragments are adjacent columns) --------------------------
    // Staging image [16 rows][64 columns] fp32 = 4 KiB per wave at `scratch`: half HALF (rows 16 HALF + [0,16)) of the fragment
    // pair (c0: columns 0-31, c1: columns 32-63).  One store instruction then writes 8 rows x 128 contiguous bytes (whole cache
    // lines; tools/ubench/store_pattern.hip: 128 MiB of such stores drain in 25.8 us against 34.2 us for 64-byte row pieces).
    template <int HALF>
    __device__ __forceinline__ void park_pair(const f32x16& c0, const f32x16& c1, int lane) const {
        float* st = scratch + ((lane >> 5) * 4) * 64 + (lane & 31);
#pragma unroll
        for (int r = 8 * HALF; r < 8 * HALF + 8; ++r) {
            const int row = (r & 3) + 8 * ((r >> 2) & 1);
            st[row * 64]      = alpha * c0[r];
            st[row * 64 + 32] = alpha * c1[r];
        }
    }
    // rows mB + [0,16), columns nB + [0,64)
    template <bool BF, int ST = 0>
    __device__ __forceinline__ void flush_pair(const GettParams& p, uint32_t mB, uint32_t nB, int lane) const {
        if (vecD && (beta == 0.f || vecC)) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int cidx = lane + 64 * it, row = cidx >> 3, piece = cidx & 7;
                const float* src = scratch + row * 64 + piece * 8;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4);
                const uint32_t m = mB + row, n = nB + 8 * piece;
                if (m < Mtot && n < Ntot) {
                    int64_t offD, offC;
                    offsets(p, m, n, offD, offC);
                    s16x8 out;
                    if (beta != 0.f) out = h_round8_with_c<BF>(v0, v1, beta, load16(C + offC, n));
                    else out = h_round8<BF>(v0, v1);
                    if constexpr (ST == 0) store16(D + offD, out, n);   // nontemporal: not read again by this kernel; keeps the operand panels in L2
                    else if constexpr (ST == 1) *(HGlbS8)(uintptr_t)(D + offD) = out;
                    else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(D + offD), "v"(out) : "memory");
                }
            }
            return;
        }
#pragma unroll 1
        for (int row = 0; row < 16; ++row) {           // element-wise form (any strides): lane = column
            const uint32_t m = mB + row, n = nB + lane;
            if (m < Mtot && n < Ntot) {
                int64_t offD, offC;
                offsets(p, m, n, offD, offC);
                float val = scratch[row * 64 + lane];
                *(HGlbU16)(uintptr_t)(D + offD) = (beta != 0.f) ? h_round16_with_c<BF>(val, beta, *(HGlbCU16)(uintptr_t)(C + offC)) : h_round16<BF>(val);
            }
        }
    }

    // ---- beta != 0 with 16-byte lanes in C and D, one chunk at a time (gett_h16w4x_kernel's pipelined beta path) ----------------------
    // chunk IT (0 .. 7) of a pass of four parked 32 x 32 fragments (the layout of flush<.., NF = 4> below): fragment IT >> 1, 16-byte
    // chunk (IT & 1) * 64 + lane of its 128.  Compile-time chunk numbers and caller-named registers: an array of chunks indexed by a
    // loop variable is a stack object whenever the unroller gives up, and scratch is paid for at every dispatch.
    template <int IT>
    __device__ __forceinline__ void chunk_coords(uint32_t mB0, uint32_t nB0, uint32_t nHi, uint32_t nLo, int lane, uint32_t& m, uint32_t& n, int& ldsFloat) const {
        constexpr int f = IT >> 1;
        const uint32_t nB = nB0 + nHi * (uint32_t)(f >> 1) + nLo * (uint32_t)(f & 1);
        const int cidx = lane + 64 * (IT & 1), row = cidx >> 2, piece = cidx & 3;
        m = mB0 + (uint32_t)row;
        n = nB + 8u * (uint32_t)piece;
        ldsFloat = f * 1024 + row * 32 + piece * 8;
    }
    template <int IT>
    __device__ __forceinline__ void load_c_chunk(const GettParams& p, uint32_t mB0, uint32_t nB0, uint32_t nHi, uint32_t nLo, int lane, s16x8& cv) const {
        uint32_t m, n;
        int at;
        chunk_coords<IT>(mB0, nB0, nHi, nLo, lane, m, n, at);
        if (m < Mtot && n < Ntot) {
            int64_t offD, offC;
            offsets(p, m, n, offD, offC);
            cv = load16(C + offC, n);
        }
    }
    template <bool BF, int IT>
    __device__ __forceinline__ void store_chunk_with_c(const GettParams& p, uint32_t mB0, uint32_t nB0, uint32_t nHi, uint32_t nLo, int lane, const s16x8& cv) const {
        uint32_t m, n;
        int at;
        chunk_coords<IT>(mB0, nB0, nHi, nLo, lane, m, n, at);
        f32x4 v0 = *reinterpret_cast<const f32x4*>(scratch + at);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(scratch + at + 4);
        if (m < Mtot && n < Ntot) {
            int64_t offD, offC;
            offsets(p, m, n, offD, offC);
            const s16x8 out = h_round8_with_c<BF>(v0, v1, beta, cv);
            store16(D + offD, out, n);
        }
    }

    // ---- four-fragment form (four-wave and streamed kernels) ------------------------------------------------------------
    // store the four parked fragments; fragment f covers rows mB + mHi (f >> 1) + mLo (f & 1) + [0, 32), columns alike
    // (base + steps, not arrays of four: a runtime-indexed array lands on the stack, and a scratch allocation is paid for at
    // every dispatch)
    // ST (measurement): 0 = nontemporal stores, 1 = plain, 2 = write-through (sc1)
    // NF: fragments parked (4; 1: fragment 0 only — the 64 x 64 kernel's one fragment per wave)
    template <bool BF, int ST = 0, int NF = 4>
    __device__ __forceinline__ void flush(const GettParams& p, uint32_t mB0, uint32_t mHi, uint32_t mLo, uint32_t nB0, uint32_t nHi, uint32_t nLo, int lane) const {
        if (vecD) {
#pragma unroll 2
            for (int it = 0; it < 2 * NF; ++it) {         // fragment it >> 1, chunks (it & 1) * 64 + lane of its 128
                const int f = it >> 1;
                const uint32_t mB = mB0 + mHi * (uint32_t)(f >> 1) + mLo * (uint32_t)(f & 1);
                const uint32_t nB = nB0 + nHi * (uint32_t)(f >> 1) + nLo * (uint32_t)(f & 1);
                const int cidx = lane + 64 * (it & 1), row = cidx >> 2, piece = cidx & 3;
                const float* src = scratch + f * 1024 + row * 32 + piece * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
                const uint32_t m = mB + row, n = nB + 8 * piece;
                if (m < Mtot && n < Ntot) {
                    int64_t offD, offC;
                    offsets(p, m, n, offD, offC);
                    const f32x4 v0 = lo, v1 = hi;
                    s16x8 out;
                    if (beta != 0.f) {
                        s16x8 cv = {0, 0, 0, 0, 0, 0, 0, 0};     // explicit elements below: nothing here may become a stack array
                        if (vecC) cv = load16(C + offC, n);
                        else {                                    // C is gathered element by element (a column past the end contributes zero)
#define CTAMD_EP_CS(E) if (n + (E) < Ntot) { int64_t oD_, oC_; offsets(p, m, n + (E), oD_, oC_); cv[E] = (short)*(HGlbCU16)(uintptr_t)(C + oC_); }
                            CTAMD_EP_CS(0) CTAMD_EP_CS(1) CTAMD_EP_CS(2) CTAMD_EP_CS(3) CTAMD_EP_CS(4) CTAMD_EP_CS(5) CTAMD_EP_CS(6) CTAMD_EP_CS(7)
#undef CTAMD_EP_CS
                        }
                        out = h_round8_with_c<BF>(v0, v1, beta, cv);
                    } else out = h_round8<BF>(v0, v1);
                    if constexpr (ST == 0) store16(D + offD, out, n);   // nontemporal: the result is not read again by this kernel
                    else if constexpr (ST == 1) *(HGlbS8)(uintptr_t)(D + offD) = out;
                    else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(D + offD), "v"(out) : "memory");
                }
            }
            return;
        }
        // element-wise form (any strides): lane = column, one row per iteration
#pragma unroll 1
        for (int it = 0; it < NF * 32; ++it) {
            const int f = it >> 5, row = it & 31;
            const uint32_t mB = mB0 + mHi * (uint32_t)(f >> 1) + mLo * (uint32_t)(f & 1);
            const uint32_t nB = nB0 + nHi * (uint32_t)(f >> 1) + nLo * (uint32_t)(f & 1);
            if (lane < 32) {
                const uint32_t m = mB + row, n = nB + lane;
                if (m < Mtot && n < Ntot) {
                    int64_t offD, offC;
                    offsets(p, m, n, offD, offC);
                    float val = scratch[f * 1024 + row * 32 + lane];
                    *(HGlbU16)(uintptr_t)(D + offD) = (beta != 0.f) ? h_round16_with_c<BF>(val, beta, *(HGlbCU16)(uintptr_t)(C + offC)) : h_round16<BF>(val);
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// One operand (A rows or B columns) of the streamed K-tile.
// ---------------------------------------------------------------------------------------------
// IL (the default kernel's B operand): half-tile h holds the 32-row stripes {64 j + 32 h + [0,32)}, j = 0..3, of the 256 rows
// instead of rows 128 h + [0,128) — the two fragments a wave owns (stripe j = wc of either half) are then ADJACENT columns of
// the output tile, and the epilogue stores whole 128-byte lines.
// SWZ (free-contiguous image only; gett_h16w4x_kernel): unit p of k-row k holds row-unit p ^ 4 (k & 3) ^ 2 ((k >> 3) & 1) — the
// 16-row fragments of the 16x16x32 MFMA take 32 bytes of a k-row per 16-lane group, and without the second term the two groups
// that a transposing read serves together (k-rows 8 g + .. and 8 (g + 1) + ..) land on the same banks.
// NH: half-tiles of 128 rows this operand stages per K-tile (2: the 256-row tiles; 1: the 128 x 128 mid-size kernel).
template <int LAY, int NW = 8, bool IL = false, int SWZ = 0, int NH = 2>
struct HOperand {
    static constexpr int kPieces = 16 / NW;   // 1-KiB pieces of a half-tile this wave stages
    // Byte offset of this lane's 16-byte unit, [half-tile][piece i of this wave], for the K-tile at k = 0 — relative to
    // `base`, the smallest such offset in the wave: a workgroup tile spans far less than 2^32 bytes whatever the size of
    // the tensor, and `base` (64 bits, wave-uniform) goes into the buffer descriptor.
    uint32_t src[NH][kPieces];
    uint64_t base;

    __device__ __forceinline__ void init(const ModeGroup& gFree, int64_t strideK0, uint32_t row0, int wave, int lane) {
        int64_t off[NH][kPieces];
        int64_t mn = INT64_MAX;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int c = wave + NW * i;                     // 1-KiB piece of the half-tile
                if constexpr (LAY == LAY_K) {
                    const int r = 8 * c + (lane >> 3), p = lane & 7;
                    const int u = p ^ ((r >> 1) & 7);
                    uint32_t row = row0 + (IL ? 64 * (r >> 5) + 32 * h + (r & 31) : 128 * h + r);
                    if (row >= gFree.total) row = gFree.total - 1;   // clamped rows feed outputs that are never stored
                    off[h][i] = (group_offset<0>(gFree, row) + 8 * u) * 2;
                } else {
                    const int kk = 4 * c + (lane >> 4), p = lane & 15;
                    const int u = p ^ (4 * ((lane >> 4) & 3)) ^ (SWZ ? 2 * ((kk >> 3) & 1) : 0);
                    uint32_t row = row0 + (IL ? 64 * (u >> 2) + 32 * h + 8 * (u & 3) : 128 * h + 8 * u);
                    // a unit past the edge is clamped to the last one that holds rows of the mode (whole when extent % 8 == 0; a ragged
                    // extent — one M / N mode, pick_h16_choice — leaves a partial last unit whose dead rows feed outputs nobody stores)
                    if (row >= gFree.total) row = (gFree.total - 1u) & ~7u;
                    off[h][i] = (group_offset<0>(gFree, row) + (int64_t)kk * strideK0) * 2;
                }
                mn = off[h][i] < mn ? off[h][i] : mn;
            }
        const int64_t mnW = (int64_t)h_uniform64((uint64_t)h_wave_min(mn));
        base = (uint64_t)mnW;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < kPieces; ++i) src[h][i] = (uint32_t)(off[h][i] - mnW);
    }

    // X: descriptor of (operand + batch offset + this wave's base + the K-tile's offset)
    template <bool PAD = true>
    __device__ __forceinline__ void issue(HRsrc X, int h, uint32_t slotByte, int wave) const {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) h_dma16<PAD>(X, src[h][i], slotByte + (uint32_t)(wave + NW * i) * 1024u);
    }
    template <bool PAD = true>
    __device__ __forceinline__ void issue_piece(HRsrc X, int h, int i, uint32_t slotByte, int wave) const {
        h_dma16<PAD>(X, src[h][i], slotByte + (uint32_t)(wave + NW * i) * 1024u);
    }
};

// Per-lane constant parts of the fragment addresses (bytes inside a half-tile slot).
//   LAY_K: offK[s], s = 16-k step 0..3, for any 32-row fragment base rb: + rb * 128
//   LAY_F: offF, depends on (rb >> 5) through the swizzle; + s * 4096 + h * 1024
__device__ __forceinline__ uint32_t h_offK(int lane, int s) {
    const int x0 = (lane >> 5) ^ ((lane >> 1) & 7);
    return (uint32_t)((lane & 31) * 128 + (((x0 ^ (2 * s)) & 7) << 4));
}
__device__ __forceinline__ uint32_t h_offF(int lane, int rbq) {
    const int g = lane >> 4, i = lane & 15;
    const int kk = 8 * (g >> 1) + (i >> 2);
    const int u = (((rbq ^ (i >> 2)) & 3) << 2) | (2 * (g & 1) + ((i >> 1) & 1));
    return (uint32_t)(kk * 256 + (u << 4) + 8 * (i & 1));
}

template <int LAY>
__device__ __forceinline__ s16x8 h_read_frag(const char* slot, int rb, int s, const uint32_t (&offK)[4], uint32_t offF) {
    if constexpr (LAY == LAY_K) {
        return *reinterpret_cast<const s16x8*>(slot + rb * 128 + offK[s]);
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

// (h_reload_params: gett_common.h reload_params)
__device__ __forceinline__ void h_reload_params(GettParams& q) { reload_params(q); }

// Wave-uniform walk over the K-tiles of the contraction: byte offsets of tile t in A and B.  Fast-K: the
// fastest contracted digit's extent is a multiple of kHBK, so a tile never straddles a digit boundary.
struct HOdometer {
    uint32_t j0, n0, j1, e1, hi;
    uint64_t offA, offB, stepA, stepB, wrapA, wrapB;      // bytes, modulo 2^64
    __device__ static __forceinline__ uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    __device__ __forceinline__ void init(const ModeGroup& gK, uint32_t k0) {
        const uint32_t E0 = gK.div[0].d;
        n0 = sgpr(E0 / kHBK);
        e1 = sgpr(gK.div[1].d);
        const uint32_t q0 = (E0 < 2) ? k0 : fast_div(k0, gK.div[0]);
        j0 = sgpr((k0 - q0 * E0) / kHBK);
        hi = sgpr((e1 < 2) ? q0 : fast_div(q0, gK.div[1]));
        j1 = sgpr(q0 - hi * e1);
        offA = h_uniform64((uint64_t)(group_offset<0>(gK, k0) * 2));
        offB = h_uniform64((uint64_t)(group_offset<1>(gK, k0) * 2));
        stepA = h_uniform64((uint64_t)((int64_t)kHBK * gK.stride[0][0] * 2));
        stepB = h_uniform64((uint64_t)((int64_t)kHBK * gK.stride[1][0] * 2));
        wrapA = h_uniform64((uint64_t)(gK.stride[0][1] * 2) - (uint64_t)(n0 - 1) * stepA);
        wrapB = h_uniform64((uint64_t)(gK.stride[1][1] * 2) - (uint64_t)(n0 - 1) * stepB);
    }
    __device__ __forceinline__ void carry(const ModeGroup& gK) {
        const uint32_t k = hi * e1 * gK.div[0].d;
        if (k < gK.total) {
            offA = h_uniform64((uint64_t)(group_offset<0>(gK, k) * 2));
            offB = h_uniform64((uint64_t)(group_offset<1>(gK, k) * 2));
        }
    }
    __device__ __forceinline__ void advance(const ModeGroup& gK) {
        const bool c0 = (j0 + 1 == n0);
        j0 = c0 ? 0u : j0 + 1;
        offA += c0 ? wrapA : stepA;
        offB += c0 ? wrapB : stepB;
        j1 += c0 ? 1u : 0u;
        if (j1 == e1) {   // carry beyond the second digit (rare)
            j1 = 0;
            hi += 1;
            carry(gK);
        }
    }
};

}  // namespace ctamd
