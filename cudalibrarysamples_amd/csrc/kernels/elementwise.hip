// elementwise.hip — HBM-bound strided copy / permutation kernels for gfx950.
//
// Replaces the closed kernels behind cutensorPermute (reference call site:
// cuTENSOR/elementwise_permute.cu:198-200, "C_{c,w,h,n} = alpha * A_{w,h,c,n}" :51-63),
// cutensorElementwiseBinaryExecute (cuTENSOR/elementwise_binary.cu:202-205) and the permutation-only
// use of cutensorReduce by the einsum helper (cuTENSOR/einsum.cu:449-450).
//
// Every tensor is walked through strides; nothing is reshaped.  The planner hands over two tile
// modes plus a linearised remainder (Ew2DParams).  Roofline: HBM; algorithmic bytes per element =
// 2 * sizeof(T) (+ sizeof(T) when a gamma*C term is read) — elementwise_permute.cu:208.
//
//   EW_TRANSPOSE  D's stride-1 mode (dim0) differs from A's stride-1 mode (dim1).  A T0 x 64 tile (T0 = 64 / 128 / 256
//                 along dim0, chosen by the planner: the width of the WRITTEN row segment is what decides the rate) is read
//                 with 16-byte lanes along dim1 (256-B contiguous segments per row), transposed 4x4 in registers, parked in
//                 LDS as [dim1][dim0] and written with 16-byte lanes along dim0 (256-B / 512-B / 1-KiB segments).  One
//                 workgroup per tile; interior tiles take an unguarded path (all loads issued before the first use); for
//                 doubly-strided transposes the tile order goes rest-first with one contiguous eighth per XCD.  16-bit data:
//                 the same with 8 x 8 register transposes on {128, 256} x {64, 128} tiles (ew_transpose_h16_wide_kernel).
//   EW_ROWCOPY    A and D share the stride-1 mode: 16-byte lanes along it, 8 dim1-rows per
//                 workgroup, no LDS.
//   EW_GENERIC    anything else (odd extents, unaligned bases, 2- and 8-byte types): one element
//                 per lane, lanes along dim0.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <cstring>

#include "launch.h"
#include "params.h"
#include "wide_elem.h"

namespace ctamd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ew_fast_div(uint32_t n, const FastDiv& d) {
    return (d.d < 2) ? n : (__umulhi(n, d.magic) >> d.shift);
}

// offsets of a rest index in A (slot 0), D (slot 1) and C (slot 2)
__device__ __forceinline__ void rest_offsets(const ModeGroup& g, uint32_t idx, int64_t& oA, int64_t& oD,
                                             int64_t& oC) {
    oA = oD = oC = 0;
#pragma unroll
    for (int i = 0; i < kMaxGroupModes; ++i) {   // padding modes: {d = 1, magic = 0, stride = 0}
        const uint32_t q = __umulhi(idx, g.div[i].magic) >> g.div[i].shift;
        const uint32_t digit = idx - q * g.div[i].d;
        oA += (int64_t)digit * g.stride[0][i];
        oD += (int64_t)digit * g.stride[1][i];
        oC += (int64_t)digit * g.stride[2][i];
        idx = q;
    }
}

// binary combiners of the element-wise family (cutensorOperator_t values; 0 = ADD)
template <typename S>
__device__ __forceinline__ S ew_comb(int op, S x, S y) {
    switch (op) {
        case 5: return x * y;                 // CUTENSOR_OP_MUL
        case 6: return x > y ? x : y;         // CUTENSOR_OP_MAX
        case 7: return x < y ? x : y;         // CUTENSOR_OP_MIN
        default: return x + y;                // CUTENSOR_OP_ADD
    }
}
__device__ __forceinline__ f32x4 ew_comb4(int op, f32x4 x, f32x4 y) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = ew_comb<float>(op, x[e], y[e]);
    return r;
}

// offset of a rest index in the second permuted operand X (explicit stride array, same digits as rest_offsets)
__device__ __forceinline__ int64_t rest_offset_x(const ModeGroup& g, const int64_t* sx, uint32_t idx) {
    int64_t o = 0;
#pragma unroll
    for (int i = 0; i < kMaxGroupModes; ++i) {
        const uint32_t q = __umulhi(idx, g.div[i].magic) >> g.div[i].shift;
        o += (int64_t)(idx - q * g.div[i].d) * sx[i];
        idx = q;
    }
    return o;
}

struct TileId { uint32_t t0, t1, rest; };
__device__ __forceinline__ TileId decode_tile(const Ew2DParams& p, uint32_t b) {
    TileId t;
    uint32_t q = ew_fast_div(b, p.divTiles0);
    t.t0 = b - q * p.tiles0;
    const uint32_t q2 = ew_fast_div(q, p.divTiles1);
    t.t1 = q - q2 * p.tiles1;
    t.rest = q2;
    return t;
}

// Tile of workgroup-loop index b under the planner's tile order (Ew2DParams::order); false = this index names no tile.
//   order 0: ids walk dim0 tiles, dim1 tiles, rest.
//   order 1 (both the rows A is read by and the rows D is written by lie a large pitch apart, e.g. the full reversal
//   A[a,b,c] -> C[c,b,a] at 2048^3, 16 MiB on both sides): ids walk rest, then dim1, then dim0, and XCD x = workgroup id % 8
//   takes the x-th eighth of that sequence, so that at any time one XCD works inside a few dim0 / dim1 tiles — a few hundred
//   distinct pages per XCD instead of every page of both tensors (fp32: 5.79 -> 6.31 TB/s, profiles/r03_transpose_sweep3_rev.jsonl;
//   the same order WITHOUT the per-XCD split is the worst: 4.14)
__device__ __forceinline__ bool ordered_tile(const Ew2DParams& p, uint32_t b, TileId& t) {
    if (p.order == 0) { t = decode_tile(p, b); return true; }
    const uint32_t id = (b & 7u) * p.idsPerXcd + (b >> 3);
    if (id >= p.nBlocks) return false;
    const uint32_t q = ew_fast_div(id, p.divRest);
    t.rest = id - q * p.rest.total;
    const uint32_t q2 = ew_fast_div(q, p.divTiles1);
    t.t1 = q - q2 * p.tiles1;
    t.t0 = q2;
    return true;
}

// ---------------------------------------------------------------------------------------------
// EW_TRANSPOSE (fp32): requires sD0 == 1, sA1 == 1, E0 % 4 == 0, E1 % 4 == 0, every other stride
// a multiple of 4 elements and 16-byte aligned bases.
// ---------------------------------------------------------------------------------------------
constexpr int TT = 64;          // tile extent along dim1 (A's contiguous mode: 256-B read segments); also the h16 kernels' edge

// T0 = tile extent along dim0 (D's contiguous mode): 64, 128 or 256 floats = 256-B / 512-B / 1-KiB written row segments.
// The width of the WRITTEN segment is what moves the 2048^3 permutation (profiles/r03_transpose_sweep*.jsonl: 64 -> 6.14,
// 128 -> 6.45, 256 -> 6.56 TB/s with one workgroup per tile; the read width and the tile order do not matter), so the
// planner takes the widest T0 the extent fills (Ew2DParams::tile0).
// HASX: second permuted operand through a second LDS tile (a separate instantiation, so that the plain permutation
// keeps its smaller footprint)
template <bool HASX, int T0>
__global__ void __launch_bounds__(256) ew_transpose_f32_kernel(const Ew2DParams p) {
    constexpr int LD = T0 + 4;                      // LDS row stride (floats)
    constexpr int RD_PASSES = T0 / 64;              // a read pass covers 64 dim0 rows (16 lane groups x 4 rows) x 64 dim1 floats
    constexpr int LPW = T0 / 4;                     // write: lanes per dim1 row
    constexpr int RPW = 256 / LPW;                  //        dim1 rows per pass
    constexpr int WR_PASSES = TT / RPW;
    __shared__ __attribute__((aligned(16))) float tile[TT * LD];   // [dim1][dim0]
    __shared__ __attribute__((aligned(16))) float tileX[HASX ? TT * LD : 4];
    const float* X = HASX ? static_cast<const float*>(p.X) : nullptr;
    const float* A = static_cast<const float*>(p.A);
    const float* C = static_cast<const float*>(p.C);
    const float* E = static_cast<const float*>(p.E);
    float*       D = static_cast<float*>(p.D);
    const int tid = threadIdx.x;

    const uint32_t nIds = p.order ? 8u * p.idsPerXcd : p.nBlocks;
    for (uint32_t b = blockIdx.x; b < nIds; b += gridDim.x) {
        TileId t;
        if (!ordered_tile(p, b, t)) continue;
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t i0 = t.t0 * T0, i1 = t.t1 * TT;   // tile origin (dim0, dim1)

        // interior tiles (all of them when the extents divide) take the unguarded path: every load of the tile is issued
        // before the first one is used, every store is a plain scaled copy
        const bool full = (i0 + T0 <= p.E0) && (i1 + TT <= p.E1);
        // ---- read: lane -> (dim1 float4 c1 = tid%16, dim0 block r0 = tid/16 [+ 64 per pass]), 4 dim0 rows each
        {
            const uint32_t c1 = i1 + 4 * (tid & 15);
            if (full) {
                const float* src = A + oA + (int64_t)(i0 + 4 * (tid >> 4)) * p.sA0 + c1;
                f32x4 in[RD_PASSES][4];
#pragma unroll
                for (int ps = 0; ps < RD_PASSES; ++ps)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        in[ps][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (int64_t)(64 * ps + r) * p.sA0));
#pragma unroll
                for (int ps = 0; ps < RD_PASSES; ++ps)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 o = {in[ps][0][j], in[ps][1][j], in[ps][2][j], in[ps][3][j]};
                        *reinterpret_cast<f32x4*>(&tile[(4 * (tid & 15) + j) * LD + 4 * (tid >> 4) + 64 * ps]) = o;
                    }
            } else {
                // edge tile: one pass at a time under per-row bounds tests.  (Round 6 tried the interior's two phases under predicates —
                // all loads of the tile first: 10 % SLOWER on 400 x 200 x 300, profiles/r06zc_*; what ragged extents cost is the row pitch,
                // 1200- and 1600-byte rows against 128-byte lines, not the rolled loop.)
#pragma unroll 1
                for (int ps = 0; ps < RD_PASSES; ++ps) {
                    const int      l0 = 4 * (tid >> 4) + 64 * ps;
                    const uint32_t r0 = i0 + l0;
                    f32x4 in[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        in[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (c1 < p.E1 && (r0 + r) < p.E0)
                            in[r] = __builtin_nontemporal_load(
                                reinterpret_cast<const f32x4*>(A + oA + (int64_t)(r0 + r) * p.sA0 + c1));
                    }
                    // 4x4 register transpose: out[j] = (in[0][j], in[1][j], in[2][j], in[3][j])
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 o = {in[0][j], in[1][j], in[2][j], in[3][j]};
                        *reinterpret_cast<f32x4*>(&tile[(4 * (tid & 15) + j) * LD + l0]) = o;
                    }
                }
            }
            if constexpr (HASX) {
                const int64_t oX = rest_offset_x(p.rest, p.restX, t.rest);
                if (full) {      // as A's interior path: every load of the tile in flight before the first one is used (round 6)
                    const float* src = X + oX + (int64_t)(i0 + 4 * (tid >> 4)) * p.sX0 + c1;
                    f32x4 in[RD_PASSES][4];
#pragma unroll
                    for (int ps = 0; ps < RD_PASSES; ++ps)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            in[ps][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (int64_t)(64 * ps + r) * p.sX0));
#pragma unroll
                    for (int ps = 0; ps < RD_PASSES; ++ps)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 o = {in[ps][0][j], in[ps][1][j], in[ps][2][j], in[ps][3][j]};
                            *reinterpret_cast<f32x4*>(&tileX[(4 * (tid & 15) + j) * LD + 4 * (tid >> 4) + 64 * ps]) = o;
                        }
                } else
#pragma unroll 1
                for (int ps = 0; ps < RD_PASSES; ++ps) {
                    const int      l0 = 4 * (tid >> 4) + 64 * ps;
                    const uint32_t r0 = i0 + l0;
                    f32x4 in[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        in[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (c1 < p.E1 && (r0 + r) < p.E0)
                            in[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(X + oX + (int64_t)(r0 + r) * p.sX0 + c1));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 o = {in[0][j], in[1][j], in[2][j], in[3][j]};
                        *reinterpret_cast<f32x4*>(&tileX[(4 * (tid & 15) + j) * LD + l0]) = o;
                    }
                }
            }
        }
        __syncthreads();
        // ---- write: lane -> (dim0 float4 c0 = tid % LPW, dim1 row = tid / LPW + RPW * pass)
        {
            const int      l0 = 4 * (tid % LPW);
            const uint32_t c0 = i0 + l0;
            if (!HASX && full && E == nullptr && C == nullptr) {
                float* dst = D + oD + (int64_t)(i1 + tid / LPW) * p.sD1 + c0;
#pragma unroll
                for (int pass = 0; pass < WR_PASSES; ++pass) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&tile[(tid / LPW + RPW * pass) * LD + l0]);
                    v *= p.alpha;
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + (int64_t)(RPW * pass) * p.sD1));
                }
            } else if (full && (C == nullptr || p.sC0 == 1)) {
                // interior tile of a binary / trinary form (round 6): the rows of C and E this lane combines with are requested for ALL
                // passes before the first one is used, no bounds tests — the rolled loop below serialises a load's latency per pass
                // (the sample's trinary form at 512 x 256 x 256: 4.4 TB/s against 5.4 for the plain permutation of the same tensor)
                const int64_t rowD = oD + (int64_t)(i1 + tid / LPW) * p.sD1 + c0;
                f32x4 cv[WR_PASSES], ev[WR_PASSES];
                if (C != nullptr) {
                    const float* cp = C + oC + (int64_t)(i1 + tid / LPW) * p.sC1 + c0;
#pragma unroll
                    for (int pass = 0; pass < WR_PASSES; ++pass)
                        cv[pass] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(cp + (int64_t)(RPW * pass) * p.sC1));
                }
                if (E != nullptr) {
#pragma unroll
                    for (int pass = 0; pass < WR_PASSES; ++pass)
                        ev[pass] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(E + rowD + (int64_t)(RPW * pass) * p.sD1));
                }
#pragma unroll
                for (int pass = 0; pass < WR_PASSES; ++pass) {
                    const int lr = tid / LPW + RPW * pass;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&tile[lr * LD + l0]);
                    v *= p.alpha;
                    if constexpr (HASX) v = ew_comb4(p.opAB, p.xi * *reinterpret_cast<const f32x4*>(&tileX[lr * LD + l0]), v);
                    if (E != nullptr) v = ew_comb4(p.opAB, p.delta * ev[pass], v);
                    if (C != nullptr) v = ew_comb4(p.opAC, v, p.gamma * cv[pass]);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(D + rowD + (int64_t)(RPW * pass) * p.sD1));
                }
            } else {
#pragma unroll 1
                for (int pass = 0; pass < WR_PASSES; ++pass) {
                    const int      lr = tid / LPW + RPW * pass;
                    const uint32_t r1 = i1 + lr;
                    if (c0 < p.E0 && r1 < p.E1) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(&tile[lr * LD + l0]);
                        v *= p.alpha;
                        if constexpr (HASX)
                            v = ew_comb4(p.opAB, p.xi * *reinterpret_cast<const f32x4*>(&tileX[lr * LD + l0]), v);
                        if (E != nullptr)
                            v = ew_comb4(p.opAB, p.delta * *reinterpret_cast<const f32x4*>(E + oD + (int64_t)r1 * p.sD1 + c0), v);
                        if (C != nullptr) {
                            const float* cp = C + oC + (int64_t)r1 * p.sC1 + (int64_t)c0 * p.sC0;
                            f32x4 c;
                            if (p.sC0 == 1) {
                                c = *reinterpret_cast<const f32x4*>(cp);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) c[e] = cp[(int64_t)e * p.sC0];
                            }
                            v = ew_comb4(p.opAC, v, p.gamma * c);
                        }
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(D + oD + (int64_t)r1 * p.sD1 + c0));
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// EW_ROWCOPY (fp32): sD0 == 1 and sA0 == 1, E0 % 4 == 0, other strides multiples of 4, aligned.
// Tile = 256 dim0 elements (64 lanes x float4) x 8 dim1 rows (4 waves x 2 rows).
// ---------------------------------------------------------------------------------------------
constexpr int RC_T0 = 256, RC_T1 = 8;

__global__ void __launch_bounds__(256) ew_rowcopy_f32_kernel(const Ew2DParams p) {
    const float* A = static_cast<const float*>(p.A);
    const float* C = static_cast<const float*>(p.C);
    const float* E = static_cast<const float*>(p.E);
    float*       D = static_cast<float*>(p.D);
    const int tid = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < p.nBlocks; b += gridDim.x) {
        const TileId t = decode_tile(p, b);
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t c0 = t.t0 * RC_T0 + 4 * (tid & 63);
        if (c0 >= p.E0) continue;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t r1 = t.t1 * RC_T1 + (tid >> 6) * 2 + r;
            if (r1 >= p.E1) continue;
            f32x4 v = __builtin_nontemporal_load(
                reinterpret_cast<const f32x4*>(A + oA + (int64_t)r1 * p.sA1 + c0));
            v *= p.alpha;
            if (E != nullptr)
                v = ew_comb4(p.opAB, p.delta * *reinterpret_cast<const f32x4*>(E + oD + (int64_t)r1 * p.sD1 + c0), v);
            if (C != nullptr) {
                const float* cp = C + oC + (int64_t)r1 * p.sC1 + (int64_t)c0 * p.sC0;
                f32x4 c;
                if (p.sC0 == 1) {
                    c = *reinterpret_cast<const f32x4*>(cp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) c[e] = cp[(int64_t)e * p.sC0];
                }
                v = ew_comb4(p.opAC, v, p.gamma * c);
            }
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(D + oD + (int64_t)r1 * p.sD1 + c0));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 16-bit data (bf16 / fp16), same two shapes: 16-byte lanes = 8 elements, arithmetic in fp32.
//   EW_ROWCOPY   tile = 512 dim0 elements (64 lanes x 8) x 8 dim1 rows
//   EW_TRANSPOSE tile = 64 x 64 elements; LDS image [dim1][dim0] with an odd row pitch (65 elements), filled and
//                drained with 2-byte LDS accesses (16 + 16 per lane and tile — a few hundred LDS cycles against
//                ~2 k cycles of HBM time for the tile's 16 KiB), HBM sees 128-byte segments on both sides
// ---------------------------------------------------------------------------------------------
typedef uint32_t u32x4e __attribute__((ext_vector_type(4)));

template <bool BF> __device__ __forceinline__ float h16_to_f32(uint16_t v) {
    if constexpr (BF) return __uint_as_float((uint32_t)v << 16);
    else return (float)__builtin_bit_cast(_Float16, v);
}
template <bool BF> __device__ __forceinline__ uint16_t f32_to_h16(float f) {
    if constexpr (BF) {
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    } else {
        return __builtin_bit_cast(uint16_t, (_Float16)f);
    }
}
template <bool BF> __device__ __forceinline__ void h16_unpack(u32x4e v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = h16_to_f32<BF>((uint16_t)(v[i] & 0xffffu)); f[2 * i + 1] = h16_to_f32<BF>((uint16_t)(v[i] >> 16)); }
}
template <bool BF> __device__ __forceinline__ u32x4e h16_pack(const float (&f)[8]) {
    u32x4e v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (uint32_t)f32_to_h16<BF>(f[2 * i]) | ((uint32_t)f32_to_h16<BF>(f[2 * i + 1]) << 16);
    return v;
}
// out = opAC(opAB(delta * E, alpha * a), gamma * C) on 8 elements starting at D-offset offD (dim0 contiguous)
template <bool BF>
__device__ __forceinline__ u32x4e h16_combine(const Ew2DParams& p, const float (&a)[8], const uint16_t* E, const uint16_t* C,
                                              int64_t offD, int64_t offC) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = p.alpha * a[i];
    if (E != nullptr) {
        float e[8];
        h16_unpack<BF>(*reinterpret_cast<const u32x4e*>(E + offD), e);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ew_comb<float>(p.opAB, p.delta * e[i], v[i]);
    }
    if (C != nullptr) {
        float c[8];
        if (p.sC0 == 1) {
            h16_unpack<BF>(*reinterpret_cast<const u32x4e*>(C + offC), c);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = h16_to_f32<BF>(C[offC + (int64_t)i * p.sC0]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ew_comb<float>(p.opAC, v[i], p.gamma * c[i]);
    }
    return h16_pack<BF>(v);
}

template <bool BF>
__global__ void __launch_bounds__(256) ew_rowcopy_h16_kernel(const Ew2DParams p) {
    const uint16_t* A = static_cast<const uint16_t*>(p.A);
    const uint16_t* C = static_cast<const uint16_t*>(p.C);
    const uint16_t* E = static_cast<const uint16_t*>(p.E);
    uint16_t*       D = static_cast<uint16_t*>(p.D);
    const int tid = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < p.nBlocks; b += gridDim.x) {
        const TileId t = decode_tile(p, b);
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t c0 = t.t0 * 512 + 8 * (tid & 63);
        if (c0 >= p.E0) continue;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t r1 = t.t1 * 8 + (tid >> 6) * 2 + r;
            if (r1 >= p.E1) continue;
            float a[8];
            h16_unpack<BF>(__builtin_nontemporal_load(reinterpret_cast<const u32x4e*>(A + oA + (int64_t)r1 * p.sA1 + c0)), a);
            const int64_t offD = oD + (int64_t)r1 * p.sD1 + c0;
            const u32x4e out = h16_combine<BF>(p, a, E, C, offD, oC + (int64_t)r1 * p.sC1 + (int64_t)c0 * p.sC0);
            __builtin_nontemporal_store(out, reinterpret_cast<u32x4e*>(D + offD));
        }
    }
}

template <bool BF>
__global__ void __launch_bounds__(256) ew_transpose_h16_kernel(const Ew2DParams p) {
    constexpr int PITCH = 65;
    __shared__ uint16_t tile[64 * PITCH];   // [dim1][dim0]
    const uint16_t* A = static_cast<const uint16_t*>(p.A);
    const uint16_t* C = static_cast<const uint16_t*>(p.C);
    const uint16_t* E = static_cast<const uint16_t*>(p.E);
    uint16_t*       D = static_cast<uint16_t*>(p.D);
    const int tid = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < p.nBlocks; b += gridDim.x) {
        const TileId t = decode_tile(p, b);
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t i0 = t.t0 * 64, i1 = t.t1 * 64;
        // read: unit u = 8 dim1 elements of one dim0 row (8 lanes cover a 128-byte segment)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = tid + 256 * k;
            const uint32_t r0 = u >> 3, c1 = 8 * (u & 7);
            u32x4e v = {0u, 0u, 0u, 0u};
            if (i0 + r0 < p.E0 && i1 + c1 < p.E1)
                v = __builtin_nontemporal_load(reinterpret_cast<const u32x4e*>(A + oA + (int64_t)(i0 + r0) * p.sA0 + i1 + c1));
#pragma unroll
            for (int j = 0; j < 8; ++j) tile[(c1 + j) * PITCH + r0] = (uint16_t)(v[j >> 1] >> (16 * (j & 1)));
        }
        __syncthreads();
        // write: unit u = 8 dim0 elements of one dim1 row
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = tid + 256 * k;
            const uint32_t lr = u >> 3, c0 = 8 * (u & 7);
            if (i0 + c0 < p.E0 && i1 + lr < p.E1) {
                float a[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = h16_to_f32<BF>(tile[lr * PITCH + c0 + j]);
                const int64_t offD = oD + (int64_t)(i1 + lr) * p.sD1 + i0 + c0;
                const u32x4e out = h16_combine<BF>(p, a, E, C, offD, oC + (int64_t)(i1 + lr) * p.sC1 + (int64_t)(i0 + c0) * p.sC0);
                __builtin_nontemporal_store(out, reinterpret_cast<u32x4e*>(D + offD));
            }
        }
        __syncthreads();
    }
}

// 16-bit transposing kernel for FULL tiles of T0 (dim0: 128 / 256 elements = 256-B / 512-B written row segments) x 64 (dim1:
// 128-B read segments); the planner selects it when the extents divide (no edge guards), the 64 x 64 kernel above otherwise.
// A lane loads one 8 x 8 block (8 dim0 rows x 16 bytes along dim1), transposes it in registers with byte permutes and parks it
// as eight 16-byte pieces of the [dim1][dim0] LDS image; the write pass reads 16-byte pieces along dim0.  alpha == 1 without
// E / C terms is a bit copy.  Same lessons as the fp32 kernel: the WRITTEN segment width is what counts, one workgroup per tile.
template <bool BF, int T0, int T1>
__global__ void __launch_bounds__(256) ew_transpose_h16_wide_kernel(const Ew2DParams p) {
    constexpr int PITCH = T0 + 8;                   // elements; rows stay 16-byte aligned
    constexpr int OCT = T1 / 8;                     // read: 16-byte octets per dim0 row
    constexpr int BROWS = 256 / OCT;                //       8-row blocks per pass
    constexpr int RP = 8 * BROWS;                   //       dim0 rows per pass
    constexpr int RD_PASSES = (T0 + RP - 1) / RP;
    constexpr int RD_LANES = (T0 >= RP) ? 256 : (T0 / 8) * OCT;   // a 128 x 64 tile keeps half the lanes busy while reading
    constexpr int LPW = T0 / 8;                     // write: lanes per dim1 row
    constexpr int RPW = 256 / LPW;
    constexpr int WR_PASSES = T1 / RPW;
    static_assert((T0 == 256 || T0 == 128) && (T1 == 64 || T1 == 128), "tiles built: {128, 256} x {64, 128}");
    __shared__ __attribute__((aligned(16))) uint16_t tile[T1 * PITCH];   // [dim1][dim0]
    const uint16_t* A = static_cast<const uint16_t*>(p.A);
    const uint16_t* C = static_cast<const uint16_t*>(p.C);
    const uint16_t* E = static_cast<const uint16_t*>(p.E);
    uint16_t*       D = static_cast<uint16_t*>(p.D);
    const int tid = threadIdx.x;
    const uint32_t nIds = p.order ? 8u * p.idsPerXcd : p.nBlocks;
    for (uint32_t b = blockIdx.x; b < nIds; b += gridDim.x) {
        TileId t;
        if (!ordered_tile(p, b, t)) continue;
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t i0 = t.t0 * T0, i1 = t.t1 * T1;
        // ---- read + 8 x 8 register transpose
        if (tid < RD_LANES) {
            const int oct = tid % OCT;                                    // dim1 octet
            const int brow = tid / OCT;                                   // block row inside a pass
#pragma unroll
            for (int ps = 0; ps < RD_PASSES; ++ps) {
                const int r0 = 8 * brow + RP * ps;                        // first dim0 row of the block
                const uint16_t* src = A + oA + (int64_t)(i0 + r0) * p.sA0 + i1 + 8 * oct;
                u32x4e v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4e*>(src + (int64_t)k * p.sA0));
                // out[j] = (v[0].e[j], ..., v[7].e[j]); element j of v[k] is half (j & 1) of word j >> 1
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    u32x4e o;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t lo = v[2 * w][j >> 1], hi = v[2 * w + 1][j >> 1];
                        o[w] = (j & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                    }
                    *reinterpret_cast<u32x4e*>(&tile[(8 * oct + j) * PITCH + r0]) = o;
                }
            }
        }
        __syncthreads();
        // ---- write: 16-byte pieces along dim0
        {
            const int l0 = 8 * (tid % LPW);
            const bool plain = (E == nullptr && C == nullptr && p.alpha == 1.0f);
            uint16_t* dst = D + oD + (int64_t)(i1 + tid / LPW) * p.sD1 + i0 + l0;
#pragma unroll
            for (int pass = 0; pass < WR_PASSES; ++pass) {
                const int lr = tid / LPW + RPW * pass;
                u32x4e v = *reinterpret_cast<const u32x4e*>(&tile[lr * PITCH + l0]);
                if (!plain) {
                    float a[8];
                    h16_unpack<BF>(v, a);
                    const int64_t offD = oD + (int64_t)(i1 + lr) * p.sD1 + i0 + l0;
                    v = h16_combine<BF>(p, a, E, C, offD, oC + (int64_t)(i1 + lr) * p.sC1 + (int64_t)(i0 + l0) * p.sC0);
                }
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4e*>(dst + (int64_t)(RPW * pass) * p.sD1));
            }
        }
        __syncthreads();
    }
}

template <bool BF>
static void launch_h16_wide(const Ew2DParams& p, unsigned grid, hipStream_t stream) {
    if (p.tile0 == 256 && p.tile1 == 128)      hipLaunchKernelGGL((ew_transpose_h16_wide_kernel<BF, 256, 128>), dim3(grid), dim3(256), 0, stream, p);
    else if (p.tile0 == 256)                   hipLaunchKernelGGL((ew_transpose_h16_wide_kernel<BF, 256, 64>), dim3(grid), dim3(256), 0, stream, p);
    else if (p.tile1 == 128)                   hipLaunchKernelGGL((ew_transpose_h16_wide_kernel<BF, 128, 128>), dim3(grid), dim3(256), 0, stream, p);
    else                                       hipLaunchKernelGGL((ew_transpose_h16_wide_kernel<BF, 128, 64>), dim3(grid), dim3(256), 0, stream, p);
}

// ---------------------------------------------------------------------------------------------
// EW_GENERIC: any strides / dtype.  Tile = 64 dim0 elements x 4 dim1 rows, one element per lane.
// ---------------------------------------------------------------------------------------------
constexpr int GN_T0 = 64, GN_T1 = 4;

template <typename T> struct EwScalar { typedef float type; };
template <> struct EwScalar<double> { typedef double type; };

template <typename T> __device__ __forceinline__ typename EwScalar<T>::type ew_load(const T* p) { return (typename EwScalar<T>::type)(*p); }
template <> __device__ __forceinline__ float ew_load<__half>(const __half* p) { return __half2float(*p); }
template <> __device__ __forceinline__ float ew_load<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void ew_store(T* p, typename EwScalar<T>::type v) { *p = (T)v; }
template <> __device__ __forceinline__ void ew_store<__half>(__half* p, float v) { *p = __float2half(v); }
template <> __device__ __forceinline__ void ew_store<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

template <typename T>
__global__ void __launch_bounds__(256) ew_generic_kernel(const Ew2DParams p) {
    typedef typename EwScalar<T>::type S;
    const T* A = static_cast<const T*>(p.A);
    const T* C = static_cast<const T*>(p.C);
    const T* E = static_cast<const T*>(p.E);
    T*       D = static_cast<T*>(p.D);
    const S alpha = sizeof(S) == 8 ? (S)p.alpha64 : (S)p.alpha;
    const S gamma = sizeof(S) == 8 ? (S)p.gamma64 : (S)p.gamma;
    const S delta = sizeof(S) == 8 ? (S)p.delta64 : (S)p.delta;
    const int tid = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < p.nBlocks; b += gridDim.x) {
        const TileId t = decode_tile(p, b);
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t c0 = t.t0 * GN_T0 + (tid & 63);
        const uint32_t r1 = t.t1 * GN_T1 + (tid >> 6);
        if (c0 >= p.E0 || r1 >= p.E1) continue;
        S v = alpha * ew_load<T>(A + oA + (int64_t)c0 * p.sA0 + (int64_t)r1 * p.sA1);
        if (E != nullptr) v = ew_comb<S>(p.opAB, delta * ew_load<T>(E + oD + (int64_t)c0 * p.sD0 + (int64_t)r1 * p.sD1), v);
        if (p.X != nullptr) {
            const S xi = sizeof(S) == 8 ? (S)p.xi64 : (S)p.xi;
            v = ew_comb<S>(p.opAB, xi * ew_load<T>(static_cast<const T*>(p.X) + rest_offset_x(p.rest, p.restX, t.rest) +
                                                  (int64_t)c0 * p.sX0 + (int64_t)r1 * p.sX1), v);
        }
        if (C != nullptr) v = ew_comb<S>(p.opAC, v, gamma * ew_load<T>(C + oC + (int64_t)c0 * p.sC0 + (int64_t)r1 * p.sC1));
        ew_store<T>(D + oD + (int64_t)c0 * p.sD0 + (int64_t)r1 * p.sD1, v);
    }
}

// ---------------------------------------------------------------------------------------------
// EW_BLOCK (round 6): D = alpha * perm(A) where the leading modes of D are the same packed set as the leading modes of A — every index
// of the other modes owns a block of blkTotal elements that is contiguous in A and in D and only permuted inside.  A workgroup loads
// blkGroup such blocks into LDS as they lie (coalesced), and writes them out in D's order (coalesced) through permuted LDS reads:
// 2 |D| bytes, no strided global access.  What the element-gather kernel above does with such a tensor: 13 MB of bf16 from
// [d = 50, c = 16, b = 4 | a] to [b, c, d | a] in 380 us (each lane of a store reads another 64-byte line); this kernel: one pass at the
// rate of a copy.  alpha == 1 moves the bits untouched.  HBM-bound.
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// EW_TRANSPOSE_ANY (round 6): D = alpha * perm(A) with D contiguous along dim0 and A along dim1 — the transposing kernels' case — at ANY
// extents, strides and base alignment (odd extents, 2-byte-aligned pointers: what the 16-byte-lane kernels refuse and the element-gather
// kernel above runs at 0.9-1.5 TB/s, each lane of a load on another 64-byte line).  A 64 x 64 tile through LDS, element by element:
// loads walk dim1 (A's contiguous mode), stores walk dim0 (D's), both coalesced; the LDS row pitch of 65 (fp32) / 66 (16-bit) elements
// keeps the column reads off a single bank.  HBM-bound: 2 |D| bytes.
// ---------------------------------------------------------------------------------------------
// HASC: the binary form D = opAC(alpha perm(A), gamma C) (elementwise_binary.cu:149-153) — C joins in the store phase, along D's contiguous
// mode; an instantiation of its own, so that the plain permutation carries no operand test in its store loop.
template <typename T, bool HASC>
__global__ void __launch_bounds__(256) ew_transpose_any_kernel(const Ew2DParams p) {
    constexpr int PITCH = sizeof(T) == 4 ? 65 : 66;
    __shared__ T lds[64 * PITCH];
    const T* A = static_cast<const T*>(p.A);
    const T* C = static_cast<const T*>(p.C);
    T*       D = static_cast<T*>(p.D);
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
    const bool raw = p.alpha == 1.0f;
    for (uint32_t b = blockIdx.x; b < p.nBlocks; b += gridDim.x) {
        const TileId t = decode_tile(p, b);
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t c0 = t.t0 * 64u, r0 = t.t1 * 64u;            // tile origin along dim0 / dim1
        const uint32_t n0 = (p.E0 - c0 < 64u) ? (p.E0 - c0) : 64u, n1 = (p.E1 - r0 < 64u) ? (p.E1 - r0) : 64u;
        // in: lane = position along dim1 (A's stride-1 mode), four dim0 positions per pass
        if ((uint32_t)lane < n1) {
            const T* src = A + oA + (int64_t)(r0 + (uint32_t)lane) + (int64_t)c0 * p.sA0;
#pragma unroll 4
            for (uint32_t i = (uint32_t)row; i < n0; i += 4u) lds[i * PITCH + lane] = src[(int64_t)i * p.sA0];
        }
        __syncthreads();
        // out: lane = position along dim0 (D's stride-1 mode), four dim1 positions per pass
        if ((uint32_t)lane < n0) {
            T* dst = D + oD + (int64_t)(c0 + (uint32_t)lane) + (int64_t)r0 * p.sD1;
            if constexpr (HASC) {
                const T* csrc = C + oC + (int64_t)(c0 + (uint32_t)lane) * p.sC0 + (int64_t)r0 * p.sC1;
#pragma unroll 4
                for (uint32_t j = (uint32_t)row; j < n1; j += 4u) {
                    const T v = lds[lane * PITCH + j];
                    ew_store<T>(dst + (int64_t)j * p.sD1, ew_comb<float>(p.opAC, p.alpha * ew_load<T>(&v), p.gamma * ew_load<T>(csrc + (int64_t)j * p.sC1)));
                }
            } else {
#pragma unroll 4
                for (uint32_t j = (uint32_t)row; j < n1; j += 4u) {
                    const T v = lds[lane * PITCH + j];
                    if (raw) dst[(int64_t)j * p.sD1] = v;
                    else ew_store<T>(dst + (int64_t)j * p.sD1, p.alpha * ew_load<T>(&v));
                }
            }
        }
        __syncthreads();
    }
}

// The same for 16-bit elements in PAIRS (Ew2DParams::tile0 == 128): even extents and strides, 4-byte-aligned bases — a lane loads two
// neighbouring elements along A's contiguous mode (one 4-byte load) and stores two along D's, a 128 x 128 tile through LDS (row pitch 130):
// the element-by-element form moves 128 bytes per wave instruction and stops at 3.3 TB/s on bf16.  Pure permutations (alpha as above).
template <typename T>
__global__ void __launch_bounds__(256) ew_transpose_any_pair_kernel(const Ew2DParams p) {
    constexpr int PITCH = 130;
    __shared__ __attribute__((aligned(4))) T lds[128 * PITCH];
    const T* A = static_cast<const T*>(p.A);
    T*       D = static_cast<T*>(p.D);
    const uint32_t lane = threadIdx.x & 63u, row = threadIdx.x >> 6;
    const bool raw = p.alpha == 1.0f;
    for (uint32_t b = blockIdx.x; b < p.nBlocks; b += gridDim.x) {
        const TileId t = decode_tile(p, b);
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t c0 = t.t0 * 128u, r0 = t.t1 * 128u;
        const uint32_t n0 = (p.E0 - c0 < 128u) ? (p.E0 - c0) : 128u, n1 = (p.E1 - r0 < 128u) ? (p.E1 - r0) : 128u;   // (even)
        if (2u * lane < n1) {
            const T* src = A + oA + (int64_t)(r0 + 2u * lane) + (int64_t)c0 * p.sA0;
#pragma unroll 4
            for (uint32_t i = row; i < n0; i += 4u)
                *reinterpret_cast<uint32_t*>(&lds[i * PITCH + 2u * lane]) = *reinterpret_cast<const uint32_t*>(src + (int64_t)i * p.sA0);
        }
        __syncthreads();
        if (2u * lane < n0) {
            T* dst = D + oD + (int64_t)(c0 + 2u * lane) + (int64_t)r0 * p.sD1;
#pragma unroll 4
            for (uint32_t j = row; j < n1; j += 4u) {
                T v0 = lds[(2u * lane) * PITCH + j], v1 = lds[(2u * lane + 1u) * PITCH + j];
                if (!raw) { T w0, w1; ew_store<T>(&w0, p.alpha * ew_load<T>(&v0)); ew_store<T>(&w1, p.alpha * ew_load<T>(&v1)); v0 = w0; v1 = w1; }
                uint16_t b0, b1;
                __builtin_memcpy(&b0, &v0, 2);
                __builtin_memcpy(&b1, &v1, 2);
                *reinterpret_cast<uint32_t*>(dst + (int64_t)j * p.sD1) = (uint32_t)b0 | ((uint32_t)b1 << 16);
            }
        }
        __syncthreads();
    }
}

// VEC: elements per 16-byte lane when the planner found 16-byte lanes on both sides (blkVec: block size, rest strides and base alignment
// multiples of it) — blocks are loaded 16 bytes per lane, and a lane gathers VEC consecutive output elements from LDS for ONE 16-byte
// store; VEC = 1: element by element (any extents / alignment).
template <typename T, int VEC>
__global__ void __launch_bounds__(256) ew_block_kernel(const Ew2DParams p) {
    // (dynamic LDS, sized to the group's blocks: a 6-KiB block leaves room for eight workgroups per CU where a static 32 KiB allowed five)
    extern __shared__ __attribute__((aligned(16))) unsigned char ew_block_lds[];
    T* const lds = reinterpret_cast<T*>(ew_block_lds);
    const T* A = static_cast<const T*>(p.A);
    T*       D = static_cast<T*>(p.D);
    const uint32_t P = p.blkTotal;
    const uint32_t r0 = blockIdx.x * p.blkGroup;
    const uint32_t nG = (p.blkRest.total - r0 < p.blkGroup) ? (p.blkRest.total - r0) : p.blkGroup;
    const int tid = threadIdx.x;
    struct alignas(16) Lane { T v[VEC]; };
    for (uint32_t g = 0; g < nG; ++g) {
        int64_t oA, oD, oC;
        rest_offsets(p.blkRest, r0 + g, oA, oD, oC);
        const T* src = A + oA;
        T* dst = lds + g * P;
        if constexpr (VEC > 1) {
            for (uint32_t e = tid * VEC; e < P; e += 256 * VEC) *reinterpret_cast<Lane*>(dst + e) = *reinterpret_cast<const Lane*>(src + e);
        } else {
            for (uint32_t e = tid; e < P; e += 256) dst[e] = src[e];
        }
    }
    __syncthreads();
    const bool raw = p.alpha == 1.0f;
    auto src_index = [&](uint32_t f) {
        uint32_t rem = f, idx = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < (int)p.blkN) {
                const uint32_t q = (i + 1 < (int)p.blkN) ? ew_fast_div(rem, p.blkDiv[i]) : 0u;
                idx += (rem - q * p.blkDiv[i].d) * p.blkSrc[i];
                rem = q;
            }
        }
        return idx;
    };
    for (uint32_t g = 0; g < nG; ++g) {
        int64_t oA, oD, oC;
        rest_offsets(p.blkRest, r0 + g, oA, oD, oC);
        const T* src = lds + g * P;
        T* dst = D + oD;
        if constexpr (VEC > 1) {
            for (uint32_t f = tid * VEC; f < P; f += 256 * VEC) {
                // the digits of f once (three divisions), then VEC - 1 increments with carry: the lane's VEC consecutive output elements
                uint32_t dig[4], rem = f, idx = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t q = (i + 1 < (int)p.blkN) ? ew_fast_div(rem, p.blkDiv[i]) : 0u;
                    dig[i] = (i < (int)p.blkN) ? rem - q * p.blkDiv[i].d : 0u;
                    idx += dig[i] * p.blkSrc[i];
                    rem = q;
                }
                Lane out;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const T v = src[idx];
                    if (raw) out.v[j] = v;
                    else { T w; ew_store<T>(&w, p.alpha * ew_load<T>(&v)); out.v[j] = w; }
                    if (j + 1 < VEC) {
                        bool carry = true;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (carry && i < (int)p.blkN) {
                                dig[i] += 1u; idx += p.blkSrc[i];
                                carry = dig[i] == p.blkDiv[i].d;
                                if (carry) { idx -= dig[i] * p.blkSrc[i]; dig[i] = 0u; }
                            }
                        }
                    }
                }
                *reinterpret_cast<Lane*>(dst + f) = out;
            }
        } else {
            for (uint32_t f = tid; f < P; f += 256) {
                const uint32_t idx = src_index(f);
                if (raw) dst[f] = src[idx];
                else ew_store<T>(dst + f, p.alpha * ew_load<T>(src + idx));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// EW_GENERIC for complex data (HIP_C_32F / HIP_C_64F): what the reference's binding runs for a unary einsum on complex tensors
// (python/einsum.h:326-343,430-441: cutensorCreateReduction + cutensorReduce with no reduced mode = a permutation;
// torch/einsum.cc:83 dispatches the complex types) and cutensorPermute / cutensorElementwiseBinaryExecute on complex tensors.
//   D = opAC(alpha * op(perm(A)), gamma * op(perm(C)))      alpha, gamma complex; op in {IDENTITY, CONJ}; opAC in {ADD, MUL}
// One (re, im) pair per lane: 8- / 16-byte loads and stores, 512 B / 1 KiB per wave along D's fastest mode.  Same tile
// decomposition as ew_generic_kernel (64 dim0 elements x 4 dim1 rows).  HBM-bound: 2 |D| bytes (+ |C|).
// ---------------------------------------------------------------------------------------------
template <typename R> struct EwCx { R re, im; };
template <typename R> __device__ __forceinline__ EwCx<R> cx_mul(EwCx<R> a, EwCx<R> b) { return EwCx<R>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <typename R> __device__ __forceinline__ EwCx<R> cx_comb(int op, EwCx<R> x, EwCx<R> y) {
    return op == 5 ? cx_mul(x, y) : EwCx<R>{x.re + y.re, x.im + y.im};     // CUTENSOR_OP_MUL, else ADD (the planner admits nothing else)
}

template <typename R>
__global__ void __launch_bounds__(256) ew_generic_cplx_kernel(const Ew2DParams p) {
    typedef EwCx<R> T;
    const T* A = static_cast<const T*>(p.A);
    const T* C = static_cast<const T*>(p.C);
    T*       D = static_cast<T*>(p.D);
    const T alpha = {(R)p.alpha64, (R)p.alphaIm}, gamma = {(R)p.gamma64, (R)p.gammaIm};
    const int tid = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < p.nBlocks; b += gridDim.x) {
        const TileId t = decode_tile(p, b);
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t c0 = t.t0 * GN_T0 + (tid & 63);
        const uint32_t r1 = t.t1 * GN_T1 + (tid >> 6);
        if (c0 >= p.E0 || r1 >= p.E1) continue;
        T a = A[oA + (int64_t)c0 * p.sA0 + (int64_t)r1 * p.sA1];
        if (p.conjA) a.im = -a.im;
        T v = cx_mul(alpha, a);
        if (C != nullptr) {
            T c = C[oC + (int64_t)c0 * p.sC0 + (int64_t)r1 * p.sC1];
            if (p.conjC) c.im = -c.im;
            v = cx_comb(p.opAC, v, cx_mul(gamma, c));
        }
        D[oD + (int64_t)c0 * p.sD0 + (int64_t)r1 * p.sD1] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// EW_TRANSPOSE / EW_ROWCOPY for 8- and 16-byte elements (fp64, complex64, complex128; round 6, wide_elem.h):
//   D = opAC(alpha * op(perm(A)), gamma * op(C))       op in {IDENTITY, CONJ}; opAC: ADD / MUL (+ MAX / MIN on fp64)
// the fp32 kernels' structure with a 16-byte lane = Tr::NV elements (2 / 2 / 1).  Transposing form: a T0 x T1 tile (T0 along dim0 = D's
// stride-1 mode: 1-KiB written row segments; T1 along dim1 = A's stride-1 mode: 256-B read segments) is read with 16-byte lanes along
// dim1, NV rows per lane, transposed NV x NV in registers, parked in LDS as [dim1][dim0] and written with 16-byte lanes along dim0.
// Planner conditions (plan_elementwise): sD0 == 1, sA1 == 1 (transposing) or sA0 == 1 (row copy), extents and every other stride
// multiples of NV, 16-byte-aligned descriptors, no E / X operand.  HBM-bound: 2 |D| bytes (+ |C|).
// ---------------------------------------------------------------------------------------------
template <class Tr>
__device__ __forceinline__ typename Tr::Acc w_ew_finish(const Ew2DParams& p, typename Tr::Acc a, const typename Tr::Elem* cp, bool hasC) {
    typename Tr::Acc v = Tr::scale(p.alpha64, p.alphaIm, a);
    if (hasC) v = Tr::apply(p.opAC == 0 ? W_OP_ADD : p.opAC, v, Tr::scale(p.gamma64, p.gammaIm, Tr::load1(cp, Tr::CX && p.conjC != 0)));
    return v;
}

template <class Tr, int T0, int T1>
__global__ void __launch_bounds__(256) ew_transpose_wide_kernel(const Ew2DParams p) {
    typedef typename Tr::Elem Elem;
    typedef typename Tr::Acc Acc;
    constexpr int NV = Tr::NV;
    constexpr int LD = T0 + NV;                     // LDS row stride (elements)
    constexpr int LPR = T1 / NV;                    // read: lanes per dim0 row
    constexpr int RPP = (256 / LPR) * NV;           //       dim0 rows per pass
    constexpr int RD_PASSES = T0 / RPP;
    constexpr int LPW = T0 / NV;                    // write: lanes per dim1 row
    constexpr int RPW = 256 / LPW;                  //        dim1 rows per pass
    constexpr int WR_PASSES = T1 / RPW;
    static_assert(T0 % RPP == 0 && T1 % RPW == 0 && 256 % LPR == 0 && 256 % LPW == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) Elem tile[T1 * LD];   // [dim1][dim0]
    const Elem* A = static_cast<const Elem*>(p.A);
    const Elem* C = static_cast<const Elem*>(p.C);
    Elem*       D = static_cast<Elem*>(p.D);
    const int tid = threadIdx.x;
    const bool conjA = Tr::CX && p.conjA != 0;
    const uint32_t nIds = p.order ? 8u * p.idsPerXcd : p.nBlocks;
    for (uint32_t b = blockIdx.x; b < nIds; b += gridDim.x) {
        TileId t;
        if (!ordered_tile(p, b, t)) continue;
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t i0 = t.t0 * T0, i1 = t.t1 * T1;
        const bool full = (i0 + T0 <= p.E0) && (i1 + T1 <= p.E1);
        {   // ---- read: lane -> (dim1 unit c1 = tid % LPR, dim0 rows r0 .. r0 + NV - 1), RD_PASSES passes
            const int      l1 = NV * (tid % LPR);
            const uint32_t c1 = i1 + l1;
            wu32x4 in[RD_PASSES][NV];
#pragma unroll
            for (int ps = 0; ps < RD_PASSES; ++ps) {
                const uint32_t r0 = i0 + NV * (tid / LPR) + RPP * ps;
#pragma unroll
                for (int r = 0; r < NV; ++r) {
                    in[ps][r] = wu32x4{0u, 0u, 0u, 0u};
                    if (full || (c1 < p.E1 && (r0 + r) < p.E0))
                        in[ps][r] = __builtin_nontemporal_load(reinterpret_cast<const wu32x4*>(A + oA + (int64_t)(r0 + r) * p.sA0 + c1));
                }
            }
#pragma unroll
            for (int ps = 0; ps < RD_PASSES; ++ps) {
                const int l0 = NV * (tid / LPR) + RPP * ps;
                // NV x NV register transpose: unit j of the output = element j of every input row
                Acc v[NV][NV];
#pragma unroll
                for (int r = 0; r < NV; ++r) Tr::unpack(in[ps][r], v[r], conjA);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    Acc o[NV];
#pragma unroll
                    for (int r = 0; r < NV; ++r) o[r] = v[r][j];
                    *reinterpret_cast<wu32x4*>(&tile[(l1 + j) * LD + l0]) = Tr::pack(o);
                }
            }
        }
        __syncthreads();
        {   // ---- write: lane -> (dim0 unit c0 = tid % LPW, dim1 row tid / LPW + RPW * pass)
            const int      l0 = NV * (tid % LPW);
            const uint32_t c0 = i0 + l0;
#pragma unroll 2
            for (int pass = 0; pass < WR_PASSES; ++pass) {
                const int      lr = tid / LPW + RPW * pass;
                const uint32_t r1 = i1 + lr;
                if (full || (c0 < p.E0 && r1 < p.E1)) {
                    Acc v[NV];
                    Tr::unpack(*reinterpret_cast<const wu32x4*>(&tile[lr * LD + l0]), v, false);
#pragma unroll
                    for (int e = 0; e < NV; ++e)
                        v[e] = w_ew_finish<Tr>(p, v[e], C + oC + (int64_t)r1 * p.sC1 + (int64_t)(c0 + e) * p.sC0, C != nullptr);
                    __builtin_nontemporal_store(Tr::pack(v), reinterpret_cast<wu32x4*>(D + oD + (int64_t)r1 * p.sD1 + c0));
                }
            }
        }
        __syncthreads();
    }
}

// row copy: A and D share the stride-1 mode — tile = 64 lanes x NV dim0 elements x 8 dim1 rows (4 waves x 2 rows), no LDS
template <class Tr>
__global__ void __launch_bounds__(256) ew_rowcopy_wide_kernel(const Ew2DParams p) {
    typedef typename Tr::Elem Elem;
    typedef typename Tr::Acc Acc;
    constexpr int NV = Tr::NV;
    const Elem* A = static_cast<const Elem*>(p.A);
    const Elem* C = static_cast<const Elem*>(p.C);
    Elem*       D = static_cast<Elem*>(p.D);
    const int tid = threadIdx.x;
    const bool conjA = Tr::CX && p.conjA != 0;
    for (uint32_t b = blockIdx.x; b < p.nBlocks; b += gridDim.x) {
        const TileId t = decode_tile(p, b);
        int64_t oA, oD, oC;
        rest_offsets(p.rest, t.rest, oA, oD, oC);
        const uint32_t c0 = t.t0 * (64u * NV) + (uint32_t)NV * (tid & 63);
        if (c0 >= p.E0) continue;
        wu32x4 raw[2];
        uint32_t r1[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            r1[r] = t.t1 * 8u + (tid >> 6) * 2 + r;
            raw[r] = wu32x4{0u, 0u, 0u, 0u};
            if (r1[r] < p.E1) raw[r] = __builtin_nontemporal_load(reinterpret_cast<const wu32x4*>(A + oA + (int64_t)r1[r] * p.sA1 + c0));
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r1[r] >= p.E1) continue;
            Acc v[NV];
            Tr::unpack(raw[r], v, conjA);
#pragma unroll
            for (int e = 0; e < NV; ++e)
                v[e] = w_ew_finish<Tr>(p, v[e], C + oC + (int64_t)r1[r] * p.sC1 + (int64_t)(c0 + e) * p.sC0, C != nullptr);
            __builtin_nontemporal_store(Tr::pack(v), reinterpret_cast<wu32x4*>(D + oD + (int64_t)r1[r] * p.sD1 + c0));
        }
    }
}

template <class Tr, int T0, int T1>
static hipError_t launch_wide_ew(const Ew2DParams& p, int variant, unsigned grid, hipStream_t stream) {
    if (p.E != nullptr || p.X != nullptr) return hipErrorInvalidValue;    // the planner never pairs these variants with a trinary operand
    if (variant == EW_TRANSPOSE) hipLaunchKernelGGL((ew_transpose_wide_kernel<Tr, T0, T1>), dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(ew_rowcopy_wide_kernel<Tr>, dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// Contiguous fill with 16-byte stores (HBM-bound: n * sizeof(T) bytes written).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct FillPattern { uint32_t w[4]; };

__global__ void __launch_bounds__(256) ew_fill_kernel(u32x4* D16, uint64_t n16, unsigned char* tail, uint32_t tailBytes, FillPattern pat) {
    const u32x4 v = {pat.w[0], pat.w[1], pat.w[2], pat.w[3]};
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        __builtin_nontemporal_store(v, D16 + i);
    if (blockIdx.x == 0 && threadIdx.x < tailBytes) tail[threadIdx.x] = (unsigned char)(pat.w[(threadIdx.x >> 2) & 3] >> (8 * (threadIdx.x & 3)));
}

static uint16_t fill_f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static uint16_t fill_f32_to_f16(float f) {   // round to nearest even, host-side
    const _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}

hipError_t launch_fill(void* D, uint64_t n, int dtype, double value, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    FillPattern pat;
    size_t es;
    switch (dtype) {
        case HIP_R_32F: { const float v = (float)value; uint32_t u; std::memcpy(&u, &v, 4); pat = FillPattern{{u, u, u, u}}; es = 4; break; }
        case HIP_R_64F: { uint64_t u; std::memcpy(&u, &value, 8); pat = FillPattern{{(uint32_t)u, (uint32_t)(u >> 32), (uint32_t)u, (uint32_t)(u >> 32)}}; es = 8; break; }
        case HIP_R_16F: { const uint32_t u = fill_f32_to_f16((float)value), w = u | (u << 16); pat = FillPattern{{w, w, w, w}}; es = 2; break; }
        case HIP_R_16BF: { const uint32_t u = fill_f32_to_bf16((float)value), w = u | (u << 16); pat = FillPattern{{w, w, w, w}}; es = 2; break; }
        // complex: the padding value is real (CUTENSOR_OPERATION_DESCRIPTOR_PADDING_VALUE is read as one scalar), imaginary part 0
        case HIP_C_32F: { const float v = (float)value; uint32_t u; std::memcpy(&u, &v, 4); pat = FillPattern{{u, 0u, u, 0u}}; es = 8; break; }
        case HIP_C_64F: { uint64_t u; std::memcpy(&u, &value, 8); pat = FillPattern{{(uint32_t)u, (uint32_t)(u >> 32), 0u, 0u}}; es = 16; break; }
        default: return hipErrorInvalidValue;
    }
    if ((reinterpret_cast<uintptr_t>(D) & 15) != 0) return hipErrorInvalidValue;   // descriptors carry >= 16-byte alignment here
    const uint64_t bytes = n * es, n16 = bytes / 16;
    const uint32_t tailBytes = (uint32_t)(bytes % 16);
    uint64_t blocks = (n16 + 255) / 256;
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(ew_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<u32x4*>(D), n16,
                       static_cast<unsigned char*>(D) + n16 * 16, tailBytes, pat);
    return hipGetLastError();
}

hipError_t launch_elementwise(const Ew2DParams& p, int variant, int dtype, hipStream_t stream) {
    if (p.nBlocks == 0) return hipSuccess;
    if (variant == EW_BLOCK) {
        // (the plan keeps the element-gather kernel's decomposition beside the block form: an attached C / E / X operand falls back to it)
        if (p.blkN >= 2 && p.C == nullptr && p.E == nullptr && p.X == nullptr && p.blkBlocks > 0) {
            const bool vec = p.blkVec != 0u && (reinterpret_cast<uintptr_t>(p.A) & 15u) == 0u && (reinterpret_cast<uintptr_t>(p.D) & 15u) == 0u;
            const size_t esz = dtype == HIP_R_32F ? 4 : 2;
            const unsigned ldsBytes = (unsigned)(((size_t)p.blkTotal * p.blkGroup * esz + 15) & ~(size_t)15);
            switch (dtype) {
                case HIP_R_32F:
                    if (vec) hipLaunchKernelGGL((ew_block_kernel<float, 4>), dim3(p.blkBlocks), dim3(256), ldsBytes, stream, p);
                    else     hipLaunchKernelGGL((ew_block_kernel<float, 1>), dim3(p.blkBlocks), dim3(256), ldsBytes, stream, p);
                    return hipGetLastError();
                case HIP_R_16F:
                    if (vec) hipLaunchKernelGGL((ew_block_kernel<__half, 8>), dim3(p.blkBlocks), dim3(256), ldsBytes, stream, p);
                    else     hipLaunchKernelGGL((ew_block_kernel<__half, 1>), dim3(p.blkBlocks), dim3(256), ldsBytes, stream, p);
                    return hipGetLastError();
                case HIP_R_16BF:
                    if (vec) hipLaunchKernelGGL((ew_block_kernel<__hip_bfloat16, 8>), dim3(p.blkBlocks), dim3(256), ldsBytes, stream, p);
                    else     hipLaunchKernelGGL((ew_block_kernel<__hip_bfloat16, 1>), dim3(p.blkBlocks), dim3(256), ldsBytes, stream, p);
                    return hipGetLastError();
                default: break;
            }
        }
        variant = EW_GENERIC;
    }
    if (variant == EW_TRANSPOSE_ANY) {
        if (p.E != nullptr || p.X != nullptr) return hipErrorInvalidValue;   // (planned for permutations and the binary form only)
        unsigned g = p.nBlocks < (1u << 22) ? p.nBlocks : (1u << 22);
        if (p.C != nullptr) {
            switch (dtype) {
                case HIP_R_32F:  hipLaunchKernelGGL((ew_transpose_any_kernel<float, true>), dim3(g), dim3(256), 0, stream, p); return hipGetLastError();
                case HIP_R_16F:  hipLaunchKernelGGL((ew_transpose_any_kernel<__half, true>), dim3(g), dim3(256), 0, stream, p); return hipGetLastError();
                case HIP_R_16BF: hipLaunchKernelGGL((ew_transpose_any_kernel<__hip_bfloat16, true>), dim3(g), dim3(256), 0, stream, p); return hipGetLastError();
                default: return hipErrorInvalidValue;
            }
        }
        if (p.tile0 == 128u && (dtype == HIP_R_16F || dtype == HIP_R_16BF)) {     // the pair form (planned for even extents / strides)
            if (((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.D)) & 3u) != 0u) return hipErrorInvalidValue;
            if (dtype == HIP_R_16F) hipLaunchKernelGGL(ew_transpose_any_pair_kernel<__half>, dim3(g), dim3(256), 0, stream, p);
            else                    hipLaunchKernelGGL(ew_transpose_any_pair_kernel<__hip_bfloat16>, dim3(g), dim3(256), 0, stream, p);
            return hipGetLastError();
        }
        switch (dtype) {
            case HIP_R_32F:  hipLaunchKernelGGL((ew_transpose_any_kernel<float, false>), dim3(g), dim3(256), 0, stream, p); return hipGetLastError();
            case HIP_R_16F:  hipLaunchKernelGGL((ew_transpose_any_kernel<__half, false>), dim3(g), dim3(256), 0, stream, p); return hipGetLastError();
            case HIP_R_16BF: hipLaunchKernelGGL((ew_transpose_any_kernel<__hip_bfloat16, false>), dim3(g), dim3(256), 0, stream, p); return hipGetLastError();
            default: return hipErrorInvalidValue;
        }
    }
    // One workgroup per tile: a workgroup that loops over tiles serialises its own read -> barrier -> write phases, fresh
    // workgroups overlap them across the CU (2048^3 permutation, 64 x 64 tiles: 5.37 TB/s with the grid capped at 32 Ki
    // workgroups, 6.09-6.14 with one workgroup per tile; profiles/r03_transpose_sweep*.jsonl).  The grid-stride loop stays
    // for tensors beyond 2^22 tiles.
    unsigned grid = p.nBlocks;
    const unsigned cap = 1u << 22;
    if (variant == EW_TRANSPOSE && p.order) grid = 8u * p.idsPerXcd;
    if (grid > cap) grid = cap;
    if (variant == EW_TRANSPOSE && dtype == HIP_R_32F) {
        if (p.X != nullptr) {
            if (p.tile0 == 128) hipLaunchKernelGGL((ew_transpose_f32_kernel<true, 128>), dim3(grid), dim3(256), 0, stream, p);
            else                hipLaunchKernelGGL((ew_transpose_f32_kernel<true, 64>), dim3(grid), dim3(256), 0, stream, p);
        } else {
            if (p.tile0 == 256)      hipLaunchKernelGGL((ew_transpose_f32_kernel<false, 256>), dim3(grid), dim3(256), 0, stream, p);
            else if (p.tile0 == 128) hipLaunchKernelGGL((ew_transpose_f32_kernel<false, 128>), dim3(grid), dim3(256), 0, stream, p);
            else                     hipLaunchKernelGGL((ew_transpose_f32_kernel<false, 64>), dim3(grid), dim3(256), 0, stream, p);
        }
    } else if (variant == EW_ROWCOPY && dtype == HIP_R_32F) {
        hipLaunchKernelGGL(ew_rowcopy_f32_kernel, dim3(grid), dim3(256), 0, stream, p);
    } else if (variant == EW_TRANSPOSE && (dtype == HIP_R_16BF || dtype == HIP_R_16F)) {
        const bool bf = dtype == HIP_R_16BF;
        if (p.tile0 > 64) {
            if (bf) launch_h16_wide<true>(p, grid, stream); else launch_h16_wide<false>(p, grid, stream);
        } else {
            if (bf) hipLaunchKernelGGL(ew_transpose_h16_kernel<true>, dim3(grid), dim3(256), 0, stream, p);
            else    hipLaunchKernelGGL(ew_transpose_h16_kernel<false>, dim3(grid), dim3(256), 0, stream, p);
        }
    } else if (variant == EW_ROWCOPY && dtype == HIP_R_16BF) {
        hipLaunchKernelGGL(ew_rowcopy_h16_kernel<true>, dim3(grid), dim3(256), 0, stream, p);
    } else if (variant == EW_ROWCOPY && dtype == HIP_R_16F) {
        hipLaunchKernelGGL(ew_rowcopy_h16_kernel<false>, dim3(grid), dim3(256), 0, stream, p);
    } else if ((variant == EW_TRANSPOSE || variant == EW_ROWCOPY) && dtype == HIP_R_64F) {
        return launch_wide_ew<WF64, 128, 32>(p, variant, grid, stream);
    } else if ((variant == EW_TRANSPOSE || variant == EW_ROWCOPY) && dtype == HIP_C_32F) {
        return launch_wide_ew<WCplx<float>, 128, 32>(p, variant, grid, stream);
    } else if ((variant == EW_TRANSPOSE || variant == EW_ROWCOPY) && dtype == HIP_C_64F) {
        return launch_wide_ew<WCplx<double>, 64, 32>(p, variant, grid, stream);
    } else if (variant == EW_GENERIC) {
        switch (dtype) {
            case HIP_R_32F:  hipLaunchKernelGGL(ew_generic_kernel<float>, dim3(grid), dim3(256), 0, stream, p); break;
            case HIP_R_64F:  hipLaunchKernelGGL(ew_generic_kernel<double>, dim3(grid), dim3(256), 0, stream, p); break;
            case HIP_R_16F:  hipLaunchKernelGGL(ew_generic_kernel<__half>, dim3(grid), dim3(256), 0, stream, p); break;
            case HIP_R_16BF: hipLaunchKernelGGL(ew_generic_kernel<__hip_bfloat16>, dim3(grid), dim3(256), 0, stream, p); break;
            case HIP_C_32F:  if (p.E != nullptr || p.X != nullptr) return hipErrorInvalidValue;
                             hipLaunchKernelGGL(ew_generic_cplx_kernel<float>, dim3(grid), dim3(256), 0, stream, p); break;
            case HIP_C_64F:  if (p.E != nullptr || p.X != nullptr) return hipErrorInvalidValue;
                             hipLaunchKernelGGL(ew_generic_cplx_kernel<double>, dim3(grid), dim3(256), 0, stream, p); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ctamd
