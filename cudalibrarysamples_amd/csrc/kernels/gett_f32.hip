// gett_f32.hip — fp32 GETT contraction kernels for gfx950 (MI355X, CDNA4).
//
// Replaces the closed device kernels behind cutensorContract for fp32 data / fp32 compute
// (reference call sites: cuTENSOR/contraction.cu:261-265, cuTENSOR/einsum.cu:334-338).
//
// Shape of the computation.  The planner (host/plan.cpp) hands over a GEMM *view* of the
// contraction: four mixed-radix mode groups L (batch), M, N, K with per-tensor strides
// (params.h).  Nothing is transposed in memory: a workgroup gathers a BM x BK tile of A and a
// BK x BN tile of B straight from the strided tensors, 16 bytes per lane along whichever
// dimension is contiguous, stages them in LDS, and feeds v_mfma_f32_16x16x4_f32.
//
//   * operand layout LAY_F ("free-contiguous"): the fastest free mode has stride 1 — lanes read
//     float4 along rows; LDS image is [k][row] (row stride = rows+4 floats so that the four
//     k-rows one MFMA touches fall in different bank halves); fragments come out with ds_read_b32.
//   * operand layout LAY_K ("K-contiguous"): the fastest contracted mode has stride 1 — lanes read
//     float4 along k (coalesced reads of the fused K mode); LDS image is [row][k] (row stride
//     BK+4); one ds_read_b128 yields the operand registers of four consecutive MFMAs.
//   * operand layout LAY_S: arbitrary strides, 4-byte gathers; LDS image as LAY_F.
//
//   Both LAY_F and LAY_K agree on the k-pairing "MFMA j of 16-step s consumes k = 16 s + 4 q + j
//   in its k-slot q = lane>>4", so any A layout combines with any B layout.
//
//   * pipeline: register-staged double buffer.  Tile t+1 is in flight in VGPRs while tile t is
//     multiplied out of LDS; one barrier per tile.  The wait counters are the compiler's.
//   * waves: WM x WN waves own disjoint (BM/WM) x (BN/WN) sub-tiles; WK > 1 additionally splits
//     each staged K-tile across waves (used for small output tiles) and sums through LDS once.
//   * split-K: blockIdx also enumerates K slices; slices write fp32 partial tiles that
//     splitk_reduce_kernel folds with alpha/beta.
//   * launch: 1-D grid, XCD-aware bijective remap so that consecutive logical tiles (which share
//     an A or B panel) sit on the same XCD and hit its private L2.
//
// Roofline: fp32 MFMA, 256 flop/clk/CU (157.3 TFLOP/s at 2.4 GHz); algorithmic flops per
// launch = 2*L*M*N*K.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "params.h"
#include "launch.h"

namespace ctamd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& d) {
    return __umulhi(n, d.magic) >> d.shift;
}

// Element offset of group index idx in tensor slot SLOT.
template <int SLOT>
__device__ __forceinline__ int64_t group_offset(const ModeGroup& g, uint32_t idx) {
    int64_t off = 0;
    const int n = g.n;
    for (int i = 0; i < n; ++i) {
        uint32_t q = 0;
        if (i + 1 < n) q = fast_div(idx, g.div[i]);
        const uint32_t digit = idx - q * g.div[i].d;
        off += (int64_t)digit * g.stride[SLOT][i];
        idx = q;
    }
    return off;
}

// Offsets of idx in the D tensor (slot SLOT of the group) and in C (explicit stride array).
template <int SLOT>
__device__ __forceinline__ void group_offset2(const ModeGroup& g, const int64_t* cstride,
                                              uint32_t idx, int64_t& offD, int64_t& offC) {
    offD = 0;
    offC = 0;
    const int n = g.n;
    for (int i = 0; i < n; ++i) {
        uint32_t q = 0;
        if (i + 1 < n) q = fast_div(idx, g.div[i]);
        const uint32_t digit = idx - q * g.div[i].d;
        offD += (int64_t)digit * g.stride[SLOT][i];
        offC += (int64_t)digit * cstride[i];
        idx = q;
    }
}

__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nBlocks) {
    // Workgroup b is dispatched to XCD b % 8 (observed, used for speed only).  Give every XCD a
    // contiguous range of logical tile ids; bijective for any nBlocks.
    const uint32_t q = nBlocks >> 3, r = nBlocks & 7u;
    const uint32_t xcd = b & 7u, i = b >> 3;
    const uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + i;
}

// ---------------------------------------------------------------------------------------------
// Tile loader for one operand.  ROWS = BM or BN, SLOT_R = slot of this tensor in its free group,
// SLOT_K = slot of this tensor in the K group.
// ---------------------------------------------------------------------------------------------
template <int LAY, int ROWS, int BK, int THREADS>
struct OperandTile {
    static constexpr int UNITS   = ROWS * BK / 4;
    static constexpr int NU      = UNITS / THREADS;
    static constexpr int RV      = ROWS / 4;   // float4 units per k-row (LAY_F / LAY_S)
    static constexpr int KV      = BK / 4;     // float4 units per row   (LAY_K)
    static constexpr int LDS_STRIDE = (LAY == LAY_K) ? (BK + 4) : (ROWS + 4);
    static constexpr int LDS_FLOATS = (LAY == LAY_K) ? ROWS * (BK + 4) : BK * (ROWS + 4);
    static constexpr int NROWOFF = (LAY == LAY_S) ? 4 * NU : NU;
    static_assert(UNITS % THREADS == 0, "tile must divide evenly over the workgroup");

    int64_t  rowOff[NROWOFF];   // element offset of this unit's row(s); -1 = out of range
    f32x4    v[NU];             // staged data

    template <int SLOT_R>
    __device__ __forceinline__ void init_rows(const ModeGroup& g, uint32_t row0, int tid) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * THREADS;
            if constexpr (LAY == LAY_K) {
                const uint32_t r = row0 + u / KV;
                rowOff[i] = (r < g.total) ? group_offset<SLOT_R>(g, r) : -1;
            } else if constexpr (LAY == LAY_F) {
                const uint32_t r = row0 + 4 * (u % RV);
                rowOff[i] = (r < g.total) ? group_offset<SLOT_R>(g, r) : -1;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t r = row0 + 4 * (u % RV) + e;
                    rowOff[4 * i + e] = (r < g.total) ? group_offset<SLOT_R>(g, r) : -1;
                }
            }
        }
    }

    // Issue the global loads of the K-tile starting at k0 (elements k0 .. k0+BK-1, clipped to kEnd).
    template <int SLOT_K>
    __device__ __forceinline__ void load(const float* __restrict__ X, const ModeGroup& gK,
                                         uint32_t k0, uint32_t kEnd, int tid) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * THREADS;
            const uint32_t k = (LAY == LAY_K) ? k0 + 4 * (u % KV) : k0 + u / RV;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (k < kEnd) {
                const int64_t offK = group_offset<SLOT_K>(gK, k);
                if constexpr (LAY == LAY_S) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int64_t ro = rowOff[4 * i + e];
                        if (ro >= 0) val[e] = X[ro + offK];
                    }
                } else {
                    const int64_t ro = rowOff[i];
                    if (ro >= 0) val = *reinterpret_cast<const f32x4*>(X + ro + offK);
                }
            }
            v[i] = val;
        }
    }

    __device__ __forceinline__ void store_lds(float* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * THREADS;
            int idx;
            if constexpr (LAY == LAY_K) idx = (u / KV) * LDS_STRIDE + 4 * (u % KV);
            else                        idx = (u / RV) * LDS_STRIDE + 4 * (u % RV);
            *reinterpret_cast<f32x4*>(lds + idx) = v[i];
        }
    }

    // Operand registers of the four MFMAs of 16-step s for the 16-row fragment starting at
    // tile-row rbase: out[j] feeds MFMA j (k = 16 s + 4 q + j, q = lane >> 4).
    __device__ static __forceinline__ f32x4 fragment(const float* lds, int rbase, int s, int lane) {
        const int i = lane & 15, q = lane >> 4;
        if constexpr (LAY == LAY_K) {
            return *reinterpret_cast<const f32x4*>(lds + (rbase + i) * LDS_STRIDE + 16 * s + 4 * q);
        } else {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = lds[(16 * s + 4 * q + j) * LDS_STRIDE + rbase + i];
            return o;
        }
    }
};

template <int BM_, int BN_, int BK_, int WM_, int WN_, int WK_, int LA_, int LB_, int MINW_>
struct GettCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, WK = WK_;
    static constexpr int LA = LA_, LB = LB_, MINW = MINW_;
    static constexpr int THREADS = 64 * WM * WN * WK;
    static constexpr int TM = BM / (WM * 16), TN = BN / (WN * 16);
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "wave sub-tile must be 16-granular");
    static_assert(BK % (16 * WK) == 0, "each wave needs whole 16-steps");
};

template <class Cfg>
__global__ void __launch_bounds__(Cfg::THREADS, Cfg::MINW) gett_f32_kernel(const GettParams p) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, WK = Cfg::WK;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, THREADS = Cfg::THREADS;
    using TileA = OperandTile<Cfg::LA, BM, BK, THREADS>;
    using TileB = OperandTile<Cfg::LB, BN, BK, THREADS>;
    constexpr int STAGE_FLOATS = TileA::LDS_FLOATS + TileB::LDS_FLOATS;
    constexpr int RED_FLOATS   = (WK > 1) ? (WK - 1) * BM * BN : 0;
    constexpr int LDS_FLOATS   = (2 * STAGE_FLOATS > RED_FLOATS) ? 2 * STAGE_FLOATS : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = (wave / WM) % WN, wk = wave / (WM * WN);

    // ---- which tile / slice / batch entry is this workgroup -------------------------------
    uint32_t id = xcd_remap(blockIdx.x, p.nBlocks);
    const uint32_t mt = id % p.tilesM; id /= p.tilesM;
    const uint32_t nt = id % p.tilesN; id /= p.tilesN;
    const uint32_t slice = id % p.splitK;
    const uint32_t l = id / p.splitK;
    const uint32_t m0 = mt * BM, n0 = nt * BN;
    const uint32_t kBegin = slice * p.kPerSlice;
    uint32_t kEnd = kBegin + p.kPerSlice;
    if (kEnd > p.gK.total) kEnd = p.gK.total;

    const float* A = static_cast<const float*>(p.A);
    const float* B = static_cast<const float*>(p.B);
    if (p.gL.n > 0) {
        A += group_offset<0>(p.gL, l);
        B += group_offset<1>(p.gL, l);
    }

    TileA ta;
    TileB tb;
    ta.template init_rows<0>(p.gM, m0, tid);
    tb.template init_rows<0>(p.gN, n0, tid);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nTiles = (kEnd > kBegin) ? (int)((kEnd - kBegin + BK - 1) / BK) : 0;

    if (nTiles > 0) {
        ta.template load<0>(A, p.gK, kBegin, kEnd, tid);
        tb.template load<1>(B, p.gK, kBegin, kEnd, tid);
        ta.store_lds(lds, tid);
        tb.store_lds(lds + TileA::LDS_FLOATS, tid);
        if (nTiles > 1) {
            ta.template load<0>(A, p.gK, kBegin + BK, kEnd, tid);
            tb.template load<1>(B, p.gK, kBegin + BK, kEnd, tid);
        }
        __syncthreads();
    }

    for (int t = 0; t < nTiles; ++t) {
        const float* la = lds + (t & 1) * STAGE_FLOATS;
        const float* lb = la + TileA::LDS_FLOATS;
#pragma unroll
        for (int ss = 0; ss < BK / (16 * WK); ++ss) {
            const int s = wk + ss * WK;
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = TileA::fragment(la, wm * (BM / WM) + 16 * i, s, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = TileB::fragment(lb, wn * (BN / WN) + 16 * j, s, lane);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nTiles) {
            float* nxt = lds + ((t + 1) & 1) * STAGE_FLOATS;
            ta.store_lds(nxt, tid);
            tb.store_lds(nxt + TileA::LDS_FLOATS, tid);
            if (t + 2 < nTiles) {
                ta.template load<0>(A, p.gK, kBegin + (uint32_t)(t + 2) * BK, kEnd, tid);
                tb.template load<1>(B, p.gK, kBegin + (uint32_t)(t + 2) * BK, kEnd, tid);
            }
        }
        __syncthreads();
    }

    // ---- fold the K-split waves of this workgroup ----------------------------------------
    if constexpr (WK > 1) {
        constexpr int PER_WAVE = TM * TN * 4 * 64;
        // the last loop barrier guarantees every wave is done reading the stage buffers
        if (wk > 0) {
            float* red = lds + ((wk - 1) * (WM * WN) + (wn * WM + wm)) * PER_WAVE;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((i * TN + j) * 4 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int w = 1; w < WK; ++w) {
            const float* red = lds + ((w - 1) * (WM * WN) + (wn * WM + wm)) * PER_WAVE;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += red[((i * TN + j) * 4 + r) * 64 + lane];
        }
    }

    // ---- epilogue --------------------------------------------------------------------------
    // MFMA C/D map: acc[i][j][r] is row 4*(lane>>4)+r, column lane&15 of fragment (i, j).
    const uint32_t Mtot = p.gM.total, Ntot = p.gN.total;
    if (p.partial != nullptr) {
        float* P = p.partial + ((size_t)slice * (p.gL.n > 0 ? p.gL.total : 1u) + l) * (size_t)Mtot * Ntot;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t m = m0 + wm * (BM / WM) + 16 * i + 4 * (lane >> 4) + r;
                if (m >= Mtot) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const uint32_t n = n0 + wn * (BN / WN) + 16 * j + (lane & 15);
                    if (n < Ntot) P[(size_t)m * Ntot + n] = acc[i][j][r];
                }
            }
        return;
    }

    const float* C = static_cast<const float*>(p.C);
    float*       D = static_cast<float*>(p.D);
    if (p.gL.n > 0) {
        int64_t oD, oC;
        group_offset2<2>(p.gL, p.cStrideL, l, oD, oC);
        D += oD;
        C += oC;
    }
    int64_t offDn[TN], offCn[TN];
    bool    okN[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const uint32_t n = n0 + wn * (BN / WN) + 16 * j + (lane & 15);
        okN[j] = n < Ntot;
        offDn[j] = 0;
        offCn[j] = 0;
        if (okN[j]) group_offset2<1>(p.gN, p.cStrideN, n, offDn[j], offCn[j]);
    }
    const float alpha = p.alpha, beta = p.beta;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t m = m0 + wm * (BM / WM) + 16 * i + 4 * (lane >> 4) + r;
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
}

// ---------------------------------------------------------------------------------------------
// Split-K second stage: D[l,m,n] = alpha * sum_s partial[s][l][m][n] + beta * C[l,m,n].
// One thread per output element, n fastest (kernel-N carries C's stride-1 mode, so both the
// partial reads and the D writes are coalesced).  HBM-bound: splitK*4 bytes read per output.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const SplitKReduceParams p) {
    const uint32_t Mtot = p.gM.total, Ntot = p.gN.total;
    const uint32_t Ltot = p.gL.n > 0 ? p.gL.total : 1u;
    const size_t   plane = (size_t)Mtot * Ntot;
    const size_t   total = plane * Ltot;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const uint32_t l = (uint32_t)(e / plane);
        const size_t   rem = e - (size_t)l * plane;
        const uint32_t m = (uint32_t)(rem / Ntot);
        const uint32_t n = (uint32_t)(rem - (size_t)m * Ntot);
        float sum = 0.f;
        const float* src = p.partial + e;
        for (uint32_t s = 0; s < p.splitK; ++s) sum += src[(size_t)s * total];
        int64_t oDl = 0, oCl = 0, oDm, oCm, oDn, oCn;
        if (p.gL.n > 0) group_offset2<2>(p.gL, p.cStrideL, l, oDl, oCl);
        group_offset2<1>(p.gM, p.cStrideM, m, oDm, oCm);
        group_offset2<1>(p.gN, p.cStrideN, n, oDn, oCn);
        float val = p.alpha * sum;
        if (p.beta != 0.f) val += p.beta * static_cast<const float*>(p.C)[oCl + oCm + oCn];
        static_cast<float*>(p.D)[oDl + oDm + oDn] = val;
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel table
// ---------------------------------------------------------------------------------------------
template <class Cfg>
static hipError_t launch_cfg(const GettParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(gett_f32_kernel<Cfg>, dim3(p.nBlocks), dim3(Cfg::THREADS), 0, stream, p);
    return hipGetLastError();
}

#define CTAMD_SHAPE_LIST(X, LA, LB)              \
    X(128, 128, 32, 2, 2, 1, LA, LB, 2)          \
    X(96, 96, 32, 2, 2, 1, LA, LB, 2)            \
    X(64, 64, 32, 2, 2, 1, LA, LB, 2)            \
    X(48, 48, 64, 1, 1, 4, LA, LB, 2)            \
    X(32, 32, 64, 1, 1, 4, LA, LB, 2)

#define CTAMD_ALL_KERNELS(X)         \
    CTAMD_SHAPE_LIST(X, LAY_F, LAY_F) \
    CTAMD_SHAPE_LIST(X, LAY_F, LAY_K) \
    CTAMD_SHAPE_LIST(X, LAY_K, LAY_F) \
    CTAMD_SHAPE_LIST(X, LAY_K, LAY_K) \
    X(64, 64, 32, 2, 2, 1, LAY_S, LAY_S, 2) \
    X(32, 32, 64, 1, 1, 4, LAY_S, LAY_S, 2)

#define CTAMD_ENTRY(bm, bn, bk, wm, wn, wk, la, lb, minw) \
    {bm, bn, bk, wm, wn, wk, la, lb, 64 * wm * wn * wk,   \
     &launch_cfg<GettCfg<bm, bn, bk, wm, wn, wk, la, lb, minw>>},

static const GettKernelInfo g_gett_f32_table[] = {CTAMD_ALL_KERNELS(CTAMD_ENTRY)};

const GettKernelInfo* gett_f32_kernels(int* count) {
    *count = (int)(sizeof(g_gett_f32_table) / sizeof(g_gett_f32_table[0]));
    return g_gett_f32_table;
}

hipError_t launch_splitk_reduce(const SplitKReduceParams& p, hipStream_t stream) {
    const uint32_t Ltot = p.gL.n > 0 ? p.gL.total : 1u;
    const size_t total = (size_t)p.gM.total * p.gN.total * Ltot;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace ctamd
