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

#include <vector>

#include "params.h"
#include "launch.h"
#include "gett_common.h"

namespace ctamd {

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

    // Element offset of this lane's row(s).  Rows beyond the tensor edge are clamped to a valid row:
    // they only feed accumulators of output rows/columns that the epilogue never stores.
    int64_t  rowOff[NROWOFF];

    template <int SLOT_R>
    __device__ __forceinline__ void init_rows(const ModeGroup& g, uint32_t row0, int tid) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * THREADS;
            if constexpr (LAY == LAY_K) {
                const uint32_t r = row0 + u / KV;
                rowOff[i] = group_offset<SLOT_R>(g, r < g.total ? r : g.total - 1);
            } else if constexpr (LAY == LAY_F) {
                const uint32_t r = row0 + 4 * (u % RV);   // extent % 4 == 0: a float4 is all in or all out
                rowOff[i] = group_offset<SLOT_R>(g, r < g.total ? r : g.total - 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t r = row0 + 4 * (u % RV) + e;
                    rowOff[4 * i + e] = group_offset<SLOT_R>(g, r < g.total ? r : g.total - 1);
                }
            }
        }
    }

    // Fast-K addressing: when the extent of the fastest K mode is a multiple of BK, a K-tile never
    // crosses a period of that mode, so element (row, k0 + kl) sits at
    //     X + [rowOff + kl * stride0]  +  offset_of(k0)
    // where the bracket is a per-lane constant (folded into rowOff by fold_k) and offset_of(k0) is
    // wave-uniform (scalar unit).  The per-tile cost is one 64-bit add per load.
    template <int SLOT_K>
    __device__ __forceinline__ void fold_k(const ModeGroup& gK, int tid) {
        const int64_t stride0 = gK.stride[SLOT_K][0];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * THREADS;
            const int64_t kl = (LAY == LAY_K) ? 4 * (u % KV) : u / RV;
            if constexpr (LAY == LAY_S) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rowOff[4 * i + e] += kl * stride0;
            } else {
                rowOff[i] += kl * stride0;
            }
        }
    }

    // Issue the global loads of the K-tile starting at k0 into v[].  Branch-free and consumer-free:
    // on the generic path k positions beyond kEnd load a clamped, valid address and are zeroed later,
    // in store_lds; on the fast-K path every tile is full and nothing ever reads a load result before
    // its LDS store, so the compiler's vmcnt waits leave the younger ring slots in flight.
    // Returns the k-validity bits of this lane's units (bit i: unit i lies inside [k0, kEnd)).
    template <int SLOT_K, bool KFAST>
    __device__ __forceinline__ uint32_t load(f32x4 (&v)[NU], const float* __restrict__ X, const ModeGroup& gK,
                                             uint32_t k0, uint32_t kEnd, int tid) const {
        constexpr uint32_t KSTEP = (LAY == LAY_K) ? 4u : 1u;   // elements one unit covers along k
        int64_t off0 = 0;
        if constexpr (KFAST) off0 = group_offset<SLOT_K>(gK, k0);   // wave-uniform
        uint32_t kMask = 0;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * THREADS;
            int64_t offK = off0;
            if constexpr (KFAST) {
                kMask |= 1u << i;
            } else {
                const uint32_t kl = (LAY == LAY_K) ? 4 * (u % KV) : u / RV;
                const bool okK = (k0 + kl) < kEnd;
                const uint32_t kc = okK ? (k0 + kl) : (kEnd - KSTEP);   // clamp into the slice
                offK = group_offset<SLOT_K>(gK, kc);
                kMask |= (okK ? 1u : 0u) << i;
            }
            if constexpr (LAY == LAY_S) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = X[rowOff[4 * i + e] + offK];
            } else {
                v[i] = *reinterpret_cast<const f32x4*>(X + rowOff[i] + offK);
            }
        }
        return kMask;
    }

    // LDS store of one staged tile; on the generic path k positions beyond the slice become zeros.
    template <bool KFAST>
    __device__ static __forceinline__ void store_lds(const f32x4 (&v)[NU], uint32_t kMask, float* lds, int tid) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + i * THREADS;
            int idx;
            if constexpr (LAY == LAY_K) idx = (u / KV) * LDS_STRIDE + 4 * (u % KV);
            else                        idx = (u / RV) * LDS_STRIDE + 4 * (u % RV);
            f32x4 val = v[i];
            if constexpr (!KFAST) {
                const bool okK = (kMask >> i) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = okK ? v[i][e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(lds + idx) = val;
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

template <int BM_, int BN_, int BK_, int WM_, int WN_, int WK_, int LA_, int LB_, int MINW_, int PF_, bool KFAST_, int ABL_ = 0, int NBUF_ = 2>
struct GettCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_, WK = WK_;
    static constexpr int LA = LA_, LB = LB_, MINW = MINW_;
    static constexpr int PF = PF_;   // K-tiles in flight in registers (prefetch distance)
    static constexpr int NBUF = NBUF_;      // LDS stage buffers: 2 = store/barrier/multiply per tile, 3 = streamed fragments
    static constexpr int ABL = ABL_;        // measurement-only ablations: 1 = no refills (LDS+MFMA), 2 = no MFMA (memory path)
    static constexpr bool KFAST = KFAST_;   // fast-K addressing (extent of the fastest K mode % BK == 0)
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
    constexpr int LDS_FLOATS   = (Cfg::NBUF * STAGE_FLOATS > RED_FLOATS) ? Cfg::NBUF * STAGE_FLOATS : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    prefetch_kernarg<(int)sizeof(GettParams)>();

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    // optional phase timestamps (diagnostics only): [0] entry, [1] ring filled / first tiles staged,
    // [2] steady loop done, [3] drain done, [4] exit (shader clock); [5]/[6] entry/exit wall clock
    unsigned long long* tlog = p.timing ? p.timing + (size_t)blockIdx.x * 16 : nullptr;
    auto stamp = [&](int slot) {
        if (tlog != nullptr && tid == 0) tlog[slot] = (slot >= 5) ? wall_clock64() : __builtin_readcyclecounter();
    };
    stamp(0);
    stamp(5);
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
    A += group_offset<0>(p.gL, l);
    B += group_offset<1>(p.gL, l);

    TileA ta;
    TileB tb;
    ta.template init_rows<0>(p.gM, m0, tid);
    tb.template init_rows<0>(p.gN, n0, tid);
    constexpr bool KFAST = Cfg::KFAST;
    if constexpr (KFAST) {
        ta.template fold_k<0>(p.gK, tid);
        tb.template fold_k<1>(p.gK, tid);
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int  nTiles = (kEnd > kBegin) ? (int)((kEnd - kBegin + BK - 1) / BK) : 0;
    constexpr int PF = Cfg::PF;

    // Register ring: PF K-tiles are in flight from HBM/L2 at any time (24-48 KB per tile), which is
    // what covers the ~2 us loaded-HBM latency at 25 GB/s per CU.  All ring indices are compile-time
    // constants (the t-loop is unrolled by PF), so the compiler's counted vmcnt waits only for the
    // oldest tile at each LDS store.
    f32x4 va[PF][TileA::NU], vb[PF][TileB::NU];
    uint32_t ma[PF], mb[PF];   // k-validity bits of each ring slot (all ones on the fast-K path)

    if constexpr (Cfg::NBUF == 3) {
        // ---------------------------------------------------------------------------------------
        // Streamed pipeline (3 LDS stage buffers).  Tile T is parked in LDS two iterations before
        // it is multiplied, so when iteration t runs, tiles t and t+1 are both resident and visible:
        // the MFMA operand fragments form one continuous, double-buffered stream (the fragments of
        // the next 16-step — possibly the first step of the next tile — are fetched while the
        // current step's MFMAs issue) and the per-tile barrier only guards buffer reuse, it is no
        // longer between an LDS store and the reads that need it.
        //   iteration t:  LDS <- ring slot of tile t+2 ; ring slot <- HBM tile t+2+PF ;
        //                 16-steps of tile t (prefetching fragments) ; barrier
        // ---------------------------------------------------------------------------------------
        static_assert(Cfg::NBUF != 3 || PF >= 2, "streamed pipeline needs two tiles staged ahead");
        constexpr int NSTEP = BK / (16 * WK);
        auto stage = [&](f32x4 (&ra)[TileA::NU], f32x4 (&rb)[TileB::NU], uint32_t& mka, uint32_t& mkb, int T, bool refill) {
            float* buf = lds + (T % 3) * STAGE_FLOATS;
            TileA::template store_lds<KFAST>(ra, mka, buf, tid);
            TileB::template store_lds<KFAST>(rb, mkb, buf + TileA::LDS_FLOATS, tid);
            if (refill && Cfg::ABL != 1) {
                mka = ta.template load<0, KFAST>(ra, A, p.gK, kBegin + (uint32_t)(T + PF) * BK, kEnd, tid);
                mkb = tb.template load<1, KFAST>(rb, B, p.gK, kBegin + (uint32_t)(T + PF) * BK, kEnd, tid);
            }
        };
        auto load_frags = [&](f32x4 (&xa)[TM], f32x4 (&xb)[TN], int T, int step) {
            const float* la = lds + (T % 3) * STAGE_FLOATS;
            const float* lb = la + TileA::LDS_FLOATS;
            const int ks = wk + step * WK;
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = TileA::fragment(la, wm * (BM / WM) + 16 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) xb[j] = TileB::fragment(lb, wn * (BN / WN) + 16 * j, ks, lane);
        };
        f32x4 fa[TM], fb[TN], na[TM], nb[TN];
        auto iteration = [&](f32x4 (&ra)[TileA::NU], f32x4 (&rb)[TileB::NU], uint32_t& mka, uint32_t& mkb, int t,
                             bool stage2, bool refill2, bool hasNext) {
            if (stage2) stage(ra, rb, mka, mkb, t + 2, refill2);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s + 1 < NSTEP) load_frags(na, nb, t, s + 1);
                else if (hasNext)  load_frags(na, nb, t + 1, 0);
                if constexpr (Cfg::ABL != 2) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = na[i];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = nb[j];
            }
            __syncthreads();
        };

        if (nTiles >= PF + 2) {
            // prologue (unconditional, so the steady loop is entered with a deterministic load queue)
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                ma[s] = ta.template load<0, KFAST>(va[s], A, p.gK, kBegin + (uint32_t)s * BK, kEnd, tid);
                mb[s] = tb.template load<1, KFAST>(vb[s], B, p.gK, kBegin + (uint32_t)s * BK, kEnd, tid);
            }
            stage(va[0], vb[0], ma[0], mb[0], 0, true);
            stage(va[1 % PF], vb[1 % PF], ma[1 % PF], mb[1 % PF], 1, true);
            __syncthreads();
            load_frags(fa, fb, 0, 0);
            const int nSteady = ((nTiles - 2 - PF) / PF) * PF;   // iterations whose refill (tile t+2+PF) exists
            stamp(1);
            for (int t0 = 0; t0 < nSteady; t0 += PF) {
#pragma unroll
                for (int s = 0; s < PF; ++s)
                    iteration(va[(s + 2) % PF], vb[(s + 2) % PF], ma[(s + 2) % PF], mb[(s + 2) % PF], t0 + s, true, true, true);
            }
            stamp(2);
            for (int t0 = nSteady; t0 < nTiles; t0 += PF) {
#pragma unroll
                for (int s = 0; s < PF; ++s) {
                    const int t = t0 + s;
                    if (t < nTiles)
                        iteration(va[(s + 2) % PF], vb[(s + 2) % PF], ma[(s + 2) % PF], mb[(s + 2) % PF], t,
                                  t + 2 < nTiles, t + 2 + PF < nTiles, t + 1 < nTiles);
                }
            }
        } else {
            // short K: plain load -> LDS -> multiply, one tile at a time
            for (int t = 0; t < nTiles; ++t) {
                ma[0] = ta.template load<0, KFAST>(va[0], A, p.gK, kBegin + (uint32_t)t * BK, kEnd, tid);
                mb[0] = tb.template load<1, KFAST>(vb[0], B, p.gK, kBegin + (uint32_t)t * BK, kEnd, tid);
                float* buf = lds + (t % 3) * STAGE_FLOATS;
                TileA::template store_lds<KFAST>(va[0], ma[0], buf, tid);
                TileB::template store_lds<KFAST>(vb[0], mb[0], buf + TileA::LDS_FLOATS, tid);
                __syncthreads();
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) {
                    load_frags(fa, fb, t, s);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
                }
                __syncthreads();
            }
        }
    } else {
    // one K-tile: LDS store of ring slot s, optional refill of the slot, barrier, MFMA block
    auto tile_step = [&](f32x4 (&ra)[TileA::NU], f32x4 (&rb)[TileB::NU], uint32_t& mka, uint32_t& mkb, int t,
                         bool refill) {
        float* buf = lds + (t & 1) * STAGE_FLOATS;
        // buf was last read by compute(t-2); every wave has passed barrier(t-1) after that
        TileA::template store_lds<KFAST>(ra, mka, buf, tid);
        TileB::template store_lds<KFAST>(rb, mkb, buf + TileA::LDS_FLOATS, tid);
        if (refill && Cfg::ABL != 1) {
            mka = ta.template load<0, KFAST>(ra, A, p.gK, kBegin + (uint32_t)(t + PF) * BK, kEnd, tid);
            mkb = tb.template load<1, KFAST>(rb, B, p.gK, kBegin + (uint32_t)(t + PF) * BK, kEnd, tid);
        }
        __syncthreads();
        const float* la = buf;
        const float* lb = buf + TileA::LDS_FLOATS;
#pragma unroll
        for (int ss = 0; ss < (Cfg::ABL == 2 ? 0 : BK / (16 * WK)); ++ss) {
            const int ks = wk + ss * WK;
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = TileA::fragment(la, wm * (BM / WM) + 16 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = TileB::fragment(lb, wn * (BN / WN) + 16 * j, ks, lane);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
    };

    if (nTiles >= PF) {
        // prologue: fill the whole ring, unconditionally — the steady loop must be entered with a
        // deterministic set of loads in flight, or the compiler's wait counts collapse to vmcnt(0)
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            ma[s] = ta.template load<0, KFAST>(va[s], A, p.gK, kBegin + (uint32_t)s * BK, kEnd, tid);
            mb[s] = tb.template load<1, KFAST>(vb[s], B, p.gK, kBegin + (uint32_t)s * BK, kEnd, tid);
        }
        // steady state: every step refills its slot unconditionally (no control flow around the loads)
        const int nSteady = ((nTiles - PF) / PF) * PF;
        stamp(1);
        for (int t0 = 0; t0 < nSteady; t0 += PF) {
#pragma unroll
            for (int s = 0; s < PF; ++s) tile_step(va[s], vb[s], ma[s], mb[s], t0 + s, true);
        }
        stamp(2);
        // drain: the last PF .. 2*PF-1 tiles
        for (int t0 = nSteady; t0 < nTiles; t0 += PF) {
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                const int t = t0 + s;
                if (t < nTiles) tile_step(va[s], vb[s], ma[s], mb[s], t, t + PF < nTiles);
            }
        }
    } else {
        // fewer K-tiles than ring slots: plain load -> store -> multiply
        for (int t = 0; t < nTiles; ++t) {
            ma[0] = ta.template load<0, KFAST>(va[0], A, p.gK, kBegin + (uint32_t)t * BK, kEnd, tid);
            mb[0] = tb.template load<1, KFAST>(vb[0], B, p.gK, kBegin + (uint32_t)t * BK, kEnd, tid);
            tile_step(va[0], vb[0], ma[0], mb[0], t, false);
        }
    }

    }

    stamp(3);
    // ---- fold the K-split waves of this workgroup ----------------------------------------
    if constexpr (WK > 1) {
        constexpr int PER_WAVE = TM * TN * 4 * 64;
        __syncthreads();   // every wave is done reading the stage buffers
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
        float* P = p.partial + ((size_t)slice * p.gL.total + l) * (size_t)Mtot * Ntot;
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
        stamp(4);
        stamp(6);
        return;
    }

    // outputs with 16-byte lanes along N: whole rows through a per-wave LDS image when the stage buffers hold one per storing wave
    // (gett_common.h, round 6)
    constexpr int IMG = gett_f32_image_floats<TN>();
    if constexpr (WM * WN * IMG <= LDS_FLOATS) {
        const GettArgPtr q = gett_arg_ptr();
        const float* Cl;
        float* Dl;
        if (gett_f32_rows_ok(q, l, Cl, Dl)) {
            __syncthreads();   // every wave is done with the stage buffers (and with the other waves' partial sums in them)
            gett_store_tile_f32_rows<TM, TN>(q, Cl, Dl, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, lds + (wn * WM + wm) * IMG);
            stamp(4);
            stamp(6);
            return;
        }
    }
    gett_store_tile_f32<TM, TN>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), l, lane);
    stamp(4);
    stamp(6);
}

// ---------------------------------------------------------------------------------------------
// Ping-pong variant for split-K dominated problems (one output tile, very deep K — the headline
// einsum 'abcd,dcbe->ae').  One 8-wave workgroup per CU, organised as two 4-wave teams that own
// alternate K-tiles of the slice.  A workgroup-wide barrier separates phases; in every phase one
// team multiplies its current tile out of its LDS buffer while the other team parks its next tile
// (register ring -> LDS) and refills the ring from HBM:
//
//      phase      0     1     2     3     4   ...
//      team 0     S0    C0    S1    C1    S2          S = stage tile, C = multiply tile
//      team 1     -     S0'   C0'   S1'   C1'
//
// Each SIMD hosts one wave of each team, so its MFMA pipe always has exactly one wave feeding it
// while the LDS-store / address / load-issue work of the other wave rides along for free.  Compared
// with two independent 4-wave workgroups per CU this keeps a single prologue and a single partial
// tile per CU (half the split-K traffic) and removes the finish-time skew between co-resident
// workgroups.  Both teams fold their accumulators through LDS at the end.
// ---------------------------------------------------------------------------------------------
template <class Cfg>
__global__ void __launch_bounds__(2 * Cfg::THREADS, 2) gett_f32_pingpong_kernel(const GettParams p) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    constexpr int WM = Cfg::WM, WN = Cfg::WN;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, TT = Cfg::THREADS;   // TT = threads per team
    constexpr int PF = Cfg::PF;
    constexpr bool KFAST = Cfg::KFAST;
    static_assert(Cfg::WK == 1, "teams replace the in-tile K split");
    using TileA = OperandTile<Cfg::LA, BM, BK, TT>;
    using TileB = OperandTile<Cfg::LB, BN, BK, TT>;
    constexpr int STAGE_FLOATS = TileA::LDS_FLOATS + TileB::LDS_FLOATS;
    constexpr int PER_WAVE = TM * TN * 4 * 64;
    constexpr int RED_FLOATS = WM * WN * PER_WAVE;
    constexpr int LDS_FLOATS = (2 * STAGE_FLOATS > RED_FLOATS) ? 2 * STAGE_FLOATS : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    prefetch_kernarg<(int)sizeof(GettParams)>();

    const int tid  = threadIdx.x;
    const int team = __builtin_amdgcn_readfirstlane(tid / TT);
    const int ttid = tid - team * TT;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(ttid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    unsigned long long* tlog = p.timing ? p.timing + (size_t)blockIdx.x * 16 : nullptr;
    auto stamp = [&](int slot) {
        if (tlog != nullptr && tid == 0) tlog[slot] = (slot >= 5) ? wall_clock64() : __builtin_readcyclecounter();
    };
    stamp(0);
    stamp(5);

    uint32_t id = xcd_remap(blockIdx.x, p.nBlocks);
    const uint32_t mt = id % p.tilesM; id /= p.tilesM;
    const uint32_t nt = id % p.tilesN; id /= p.tilesN;
    const uint32_t slice = id % p.splitK;
    const uint32_t l = id / p.splitK;
    const uint32_t m0 = mt * BM, n0 = nt * BN;
    const uint32_t kBegin = slice * p.kPerSlice;
    uint32_t kEnd = kBegin + p.kPerSlice;
    if (kEnd > p.gK.total) kEnd = p.gK.total;

    const float* A = static_cast<const float*>(p.A) + group_offset<0>(p.gL, l);
    const float* B = static_cast<const float*>(p.B) + group_offset<1>(p.gL, l);

    TileA ta;
    TileB tb;
    ta.template init_rows<0>(p.gM, m0, ttid);
    tb.template init_rows<0>(p.gN, n0, ttid);
    if constexpr (KFAST) {
        ta.template fold_k<0>(p.gK, ttid);
        tb.template fold_k<1>(p.gK, ttid);
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nTiles = (kEnd > kBegin) ? (int)((kEnd - kBegin + BK - 1) / BK) : 0;
    const int nMine = (nTiles - team + 1) / 2;      // this team owns tiles team, team + 2, ...
    const int nPhases = 2 * ((nTiles + 1) / 2) + 1;   // barriers every wave must execute
    float* buf = lds + team * STAGE_FLOATS;
    auto tile_k = [&](int i) { return kBegin + (uint32_t)(2 * i + team) * BK; };

    f32x4 va[PF][TileA::NU], vb[PF][TileB::NU];
    uint32_t ma[PF], mb[PF];
    int barriers = 0;

    auto stage = [&](f32x4 (&ra)[TileA::NU], f32x4 (&rb)[TileB::NU], uint32_t& mka, uint32_t& mkb, int i, bool refill) {
        TileA::template store_lds<KFAST>(ra, mka, buf, ttid);
        TileB::template store_lds<KFAST>(rb, mkb, buf + TileA::LDS_FLOATS, ttid);
        if (refill) {
            mka = ta.template load<0, KFAST>(ra, A, p.gK, tile_k(i + PF), kEnd, ttid);
            mkb = tb.template load<1, KFAST>(rb, B, p.gK, tile_k(i + PF), kEnd, ttid);
        }
    };
    auto multiply = [&]() {
        const float* la = buf;
        const float* lb = buf + TileA::LDS_FLOATS;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = TileA::fragment(la, wm * (BM / WM) + 16 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = TileB::fragment(lb, wn * (BN / WN) + 16 * j, ks, lane);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
    };
    // one owned tile: stage | barrier | multiply | barrier
    auto step = [&](f32x4 (&ra)[TileA::NU], f32x4 (&rb)[TileB::NU], uint32_t& mka, uint32_t& mkb, int i, bool refill) {
        stage(ra, rb, mka, mkb, i, refill);
        __syncthreads();
        multiply();
        __syncthreads();
        barriers += 2;
    };

    if (team == 1) {   // team 1 runs one phase behind team 0
        __syncthreads();
        barriers += 1;
    }
    if (nMine >= PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            ma[s] = ta.template load<0, KFAST>(va[s], A, p.gK, tile_k(s), kEnd, ttid);
            mb[s] = tb.template load<1, KFAST>(vb[s], B, p.gK, tile_k(s), kEnd, ttid);
        }
        const int nSteady = ((nMine - PF) / PF) * PF;
        stamp(1);
        for (int i0 = 0; i0 < nSteady; i0 += PF) {
#pragma unroll
            for (int s = 0; s < PF; ++s) step(va[s], vb[s], ma[s], mb[s], i0 + s, true);
        }
        stamp(2);
        for (int i0 = nSteady; i0 < nMine; i0 += PF) {
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                const int i = i0 + s;
                if (i < nMine) step(va[s], vb[s], ma[s], mb[s], i, i + PF < nMine);
            }
        }
    } else {
        for (int i = 0; i < nMine; ++i) {
            ma[0] = ta.template load<0, KFAST>(va[0], A, p.gK, tile_k(i), kEnd, ttid);
            mb[0] = tb.template load<1, KFAST>(vb[0], B, p.gK, tile_k(i), kEnd, ttid);
            step(va[0], vb[0], ma[0], mb[0], i, false);
        }
    }
    for (; barriers < nPhases; ++barriers) __syncthreads();   // both teams meet the same barrier count
    stamp(3);

    // ---- fold team 1 into team 0 (all stage buffers are idle after the last barrier) --------------
    if (team == 1) {
        float* red = lds + (wn * WM + wm) * PER_WAVE;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((i * TN + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (team == 1) return;
    {
        const float* red = lds + (wn * WM + wm) * PER_WAVE;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += red[((i * TN + j) * 4 + r) * 64 + lane];
    }

    // ---- epilogue (same as gett_f32_kernel) -----------------------------------------------------------
    const uint32_t Mtot = p.gM.total, Ntot = p.gN.total;
    if (p.partial != nullptr) {
        float* P = p.partial + ((size_t)slice * p.gL.total + l) * (size_t)Mtot * Ntot;
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
        stamp(4);
        stamp(6);
        return;
    }
    // outputs with 16-byte lanes along N: whole rows through a per-wave LDS image when the stage buffers hold one per storing wave
    // (gett_common.h, round 6)
    constexpr int IMG = gett_f32_image_floats<TN>();
    if constexpr (WM * WN * IMG <= LDS_FLOATS) {
        const GettArgPtr q = gett_arg_ptr();
        const float* Cl;
        float* Dl;
        if (gett_f32_rows_ok(q, l, Cl, Dl)) {
            __syncthreads();   // every wave is done with the stage buffers (and with the other waves' partial sums in them)
            gett_store_tile_f32_rows<TM, TN>(q, Cl, Dl, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, lds + (wn * WM + wm) * IMG);
            stamp(4);
            stamp(6);
            return;
        }
    }
    gett_store_tile_f32<TM, TN>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), l, lane);
    stamp(4);
    stamp(6);
}

// ---------------------------------------------------------------------------------------------
// Split-K second stage: D[l,m,n] = alpha * sum_s partial[s][l][m][n] + beta * C[l,m,n].
// One thread per output element, n fastest (kernel-N carries C's stride-1 mode, so both the
// partial reads and the D writes are coalesced).  HBM-bound: splitK*4 bytes read per output.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) splitk_reduce_kernel(const SplitKReduceParams p) {
    // 512 lanes = 32 consecutive outputs x 16 slice groups: each lane sums every 16th slice of its
    // output (128-B coalesced rows of the partial buffer), the 16 groups meet in LDS.
    __shared__ float red[16][33];
    const uint32_t Mtot = p.gM.total, Ntot = p.gN.total;
    const uint32_t Ltot = p.gL.total;
    const size_t   plane = (size_t)Mtot * Ntot;
    const size_t   total = plane * Ltot;
    const int      o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t   e = (size_t)blockIdx.x * 32 + o;
    float sum = 0.f;
    if (e < total) {
        const float* src = p.partial + e;
        uint32_t s = g;
        for (; s + 48 < p.splitK; s += 64) {
            const float x0 = src[(size_t)s * total], x1 = src[(size_t)(s + 16) * total];
            const float x2 = src[(size_t)(s + 32) * total], x3 = src[(size_t)(s + 48) * total];
            sum += (x0 + x1) + (x2 + x3);
        }
        for (; s < p.splitK; s += 16) sum += src[(size_t)s * total];
    }
    red[g][o] = sum;
    __syncthreads();
    if (g != 0 || e >= total) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) sum += red[k][o];
    const uint32_t l = (uint32_t)(e / plane);
    const size_t   rem = e - (size_t)l * plane;
    const uint32_t m = (uint32_t)(rem / Ntot);
    const uint32_t n = (uint32_t)(rem - (size_t)m * Ntot);
    int64_t oDl = 0, oCl = 0, oDm, oCm, oDn, oCn;
    group_offset2<2>(p.gL, p.cStrideL, l, oDl, oCl);
    group_offset2<1>(p.gM, p.cStrideM, m, oDm, oCm);
    group_offset2<1>(p.gN, p.cStrideN, n, oDn, oCn);
    float val = p.alpha * sum;
    if (p.outType == 0) {
        if (p.beta != 0.f) val += p.beta * static_cast<const float*>(p.C)[oCl + oCm + oCn];
        static_cast<float*>(p.D)[oDl + oDm + oDn] = val;
    } else {   // 16-bit data of the gett_h16 kernels: fp32 partials, one rounding of the result
        const bool bf = p.outType == 1;
        if (p.beta != 0.f) {
            const uint16_t c = static_cast<const uint16_t*>(p.C)[oCl + oCm + oCn];
            val += p.beta * (bf ? __uint_as_float((uint32_t)c << 16) : (float)__builtin_bit_cast(_Float16, c));
        }
        uint16_t out;
        if (bf) {
            uint32_t u = __float_as_uint(val);
            if ((u & 0x7fffffffu) > 0x7f800000u) out = (uint16_t)((u >> 16) | 0x40u);
            else { u += 0x7fffu + ((u >> 16) & 1u); out = (uint16_t)(u >> 16); }
        } else {
            out = __builtin_bit_cast(uint16_t, (_Float16)val);
        }
        static_cast<uint16_t*>(p.D)[oDl + oDm + oDn] = out;
    }
}

// The same fold for FEW slices over MANY outputs (mid-size 16-bit contractions: 2048^3 bf16 runs 64 tiles x 4 slices): one lane per
// four consecutive n, the slices summed in sequence — 16-byte partial reads, one 16- / 8-byte store where four consecutive n are
// contiguous in D (and C).  The kernel above gives sixteen lanes to every output for the slices; with 4 slices twelve of them idle,
// and 2048^2 outputs took 171 us where this form takes the time of the traffic.
__global__ void __launch_bounds__(256) splitk_reduce_wide_kernel(const SplitKReduceParams p) {
    const uint32_t Mtot = p.gM.total, Ntot = p.gN.total;      // Ntot % 4 == 0 (launch_splitk_reduce)
    const size_t   plane = (size_t)Mtot * Ntot;
    const size_t   total = plane * p.gL.total;
    const size_t   e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= total) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(p.partial + e);
    const size_t stride4 = total / 4;
    f32x4 sum = src[0];
    for (uint32_t s = 1; s < p.splitK; ++s) sum += src[(size_t)s * stride4];
    const uint32_t l = (uint32_t)(e / plane);
    const size_t   rem = e - (size_t)l * plane;
    const uint32_t m = (uint32_t)(rem / Ntot);
    const uint32_t n = (uint32_t)(rem - (size_t)m * Ntot);
    int64_t oDl = 0, oCl = 0, oDm, oCm, oDn, oCn;
    group_offset2<2>(p.gL, p.cStrideL, l, oDl, oCl);
    group_offset2<1>(p.gM, p.cStrideM, m, oDm, oCm);
    group_offset2<1>(p.gN, p.cStrideN, n, oDn, oCn);
    const int64_t oD = oDl + oDm + oDn, oC = oCl + oCm + oCn;
    // four consecutive n stay inside the fastest N mode and are contiguous in D / C
    const bool inMode = (p.gN.div[0].d % 4u) == 0u;
    const bool vecD = inMode && p.gN.stride[1][0] == 1, vecC = inMode && p.cStrideN[0] == 1;
    float val[4] = {p.alpha * sum[0], p.alpha * sum[1], p.alpha * sum[2], p.alpha * sum[3]};
    int64_t dD[4] = {0, 1, 2, 3}, dC[4] = {0, 1, 2, 3};
    if (!vecD || (p.beta != 0.f && !vecC)) {
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            int64_t a, b;
            group_offset2<1>(p.gN, p.cStrideN, n + i, a, b);
            dD[i] = a - oDn;
            dC[i] = b - oCn;
        }
    }
    if (p.outType == 0) {
        float* D = static_cast<float*>(p.D) + oD;
        if (p.beta != 0.f) {
            const float* C = static_cast<const float*>(p.C) + oC;
#pragma unroll
            for (int i = 0; i < 4; ++i) val[i] += p.beta * C[dC[i]];
        }
        if (vecD && (reinterpret_cast<uintptr_t>(D) & 15u) == 0u) *reinterpret_cast<f32x4*>(D) = f32x4{val[0], val[1], val[2], val[3]};
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) D[dD[i]] = val[i];
        }
        return;
    }
    const bool bf = p.outType == 1;
    uint16_t* D = static_cast<uint16_t*>(p.D) + oD;
    if (p.beta != 0.f) {
        const uint16_t* C = static_cast<const uint16_t*>(p.C) + oC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint16_t c = C[dC[i]];
            val[i] += p.beta * (bf ? __uint_as_float((uint32_t)c << 16) : (float)__builtin_bit_cast(_Float16, c));
        }
    }
    uint16_t out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (bf) {
            uint32_t u = __float_as_uint(val[i]);
            if ((u & 0x7fffffffu) > 0x7f800000u) out[i] = (uint16_t)((u >> 16) | 0x40u);
            else { u += 0x7fffu + ((u >> 16) & 1u); out[i] = (uint16_t)(u >> 16); }
        } else {
            out[i] = __builtin_bit_cast(uint16_t, (_Float16)val[i]);
        }
    }
    if (vecD && (reinterpret_cast<uintptr_t>(D) & 7u) == 0u) {
        typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u16x4*>(D) = u16x4{out[0], out[1], out[2], out[3]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) D[dD[i]] = out[i];
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

template <class Cfg>
static hipError_t launch_pingpong(const GettParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(gett_f32_pingpong_kernel<Cfg>, dim3(p.nBlocks), dim3(2 * Cfg::THREADS), 0, stream, p);
    return hipGetLastError();
}

// X(bm, bn, bk, wm, wn, wk, layA, layB, min waves/SIMD, prefetch depth, fast-K)
#define CTAMD_SHAPE_LIST(X, LA, LB)                    \
    X(128, 128, 32, 2, 2, 1, LA, LB, 2, 2, true)       \
    X(96, 96, 32, 2, 2, 1, LA, LB, 2, 3, true)         \
    X(64, 64, 32, 2, 2, 1, LA, LB, 2, 3, true)         \
    X(48, 48, 64, 1, 1, 4, LA, LB, 2, 3, true)         \
    X(32, 32, 64, 1, 1, 4, LA, LB, 2, 3, true)         \
    X(64, 64, 32, 2, 2, 1, LA, LB, 2, 2, false)        \
    X(32, 32, 64, 1, 1, 4, LA, LB, 2, 2, false)

#define CTAMD_ALL_KERNELS(X)         \
    CTAMD_SHAPE_LIST(X, LAY_F, LAY_F) \
    CTAMD_SHAPE_LIST(X, LAY_F, LAY_K) \
    CTAMD_SHAPE_LIST(X, LAY_K, LAY_F) \
    CTAMD_SHAPE_LIST(X, LAY_K, LAY_K) \
    X(64, 64, 32, 2, 2, 1, LAY_S, LAY_S, 2, 2, false) \
    X(32, 32, 64, 1, 1, 4, LAY_S, LAY_S, 2, 2, false) \
    /* experimental variants of the headline shapes (prefetch depth) */ \
    X(96, 96, 32, 2, 2, 1, LAY_K, LAY_F, 2, 2, true) \
    X(96, 96, 32, 2, 2, 1, LAY_K, LAY_F, 2, 4, true) \
    X(48, 48, 64, 1, 1, 4, LAY_K, LAY_F, 2, 2, true) \
    X(48, 48, 64, 1, 1, 4, LAY_K, LAY_F, 2, 4, true) \
    X(128, 128, 32, 2, 2, 1, LAY_F, LAY_F, 2, 1, true) \
    X(128, 128, 32, 2, 2, 1, LAY_F, LAY_F, 2, 3, true)

// streamed (3-buffer) pipeline variants: X3(bm, bn, bk, wm, wn, wk, layA, layB, minw, pf)
#define CTAMD_STREAMED_KERNELS(X3)                   \
    X3(96, 96, 32, 2, 2, 1, LAY_K, LAY_F, 2, 2)      \
    X3(96, 96, 32, 2, 2, 1, LAY_K, LAY_F, 2, 3)      \
    X3(96, 96, 32, 2, 2, 1, LAY_K, LAY_F, 1, 4)      \
    X3(48, 48, 64, 1, 1, 4, LAY_K, LAY_F, 2, 2)      \
    X3(48, 48, 64, 1, 1, 4, LAY_K, LAY_F, 2, 3)      \
    X3(128, 128, 32, 2, 2, 1, LAY_F, LAY_F, 1, 2)    \
    X3(96, 96, 32, 2, 2, 1, LAY_F, LAY_F, 2, 2)      \
    X3(128, 128, 16, 2, 2, 1, LAY_F, LAY_F, 2, 3)

#define CTAMD_ENTRY(bm, bn, bk, wm, wn, wk, la, lb, minw, pf, kfast) \
    {bm, bn, bk, wm, wn, wk, la, lb, 64 * wm * wn * wk, pf, kfast ? 1 : 0, 0, \
     &launch_cfg<GettCfg<bm, bn, bk, wm, wn, wk, la, lb, minw, pf, kfast>>},
#define CTAMD_ENTRY3(bm, bn, bk, wm, wn, wk, la, lb, minw, pf) \
    {bm, bn, bk, wm, wn, wk, la, lb, 64 * wm * wn * wk, pf, 1, 0, \
     &launch_cfg<GettCfg<bm, bn, bk, wm, wn, wk, la, lb, minw, pf, true, 0, 3>>},
// two-team ping-pong kernels: XP(bm, bn, bk, wm, wn, layA, layB, pf, kfast)
#define CTAMD_PINGPONG_KERNELS(XP)             \
    XP(96, 96, 32, 2, 2, LAY_K, LAY_F, 2, true)  \
    XP(96, 96, 32, 2, 2, LAY_K, LAY_F, 3, true)  \
    XP(96, 96, 32, 2, 2, LAY_F, LAY_F, 2, true)  \
    XP(96, 96, 32, 2, 2, LAY_F, LAY_K, 2, true)  \
    XP(96, 96, 32, 2, 2, LAY_K, LAY_K, 2, true)  \
    XP(64, 64, 32, 2, 2, LAY_K, LAY_F, 2, false)
#define CTAMD_ENTRYP(bm, bn, bk, wm, wn, la, lb, pf, kfast) \
    {bm, bn, bk, wm, wn, 1, la, lb, 2 * 64 * wm * wn, pf, kfast ? 1 : 0, 0, \
     &launch_pingpong<GettCfg<bm, bn, bk, wm, wn, 1, la, lb, 2, pf, kfast>>},
// measurement-only ablations of the headline kernel (never ranked unless CUTENSOR_AMD_ABLATION is set)
#define CTAMD_ABL_ENTRY(bm, bn, bk, wm, wn, wk, la, lb, minw, pf, abl) \
    {bm, bn, bk, wm, wn, wk, la, lb, 64 * wm * wn * wk, pf, 1, abl, \
     &launch_cfg<GettCfg<bm, bn, bk, wm, wn, wk, la, lb, minw, pf, true, abl>>},

static const GettKernelInfo g_gett_f32_table[] = {
    CTAMD_ALL_KERNELS(CTAMD_ENTRY)
    CTAMD_STREAMED_KERNELS(CTAMD_ENTRY3)
    CTAMD_PINGPONG_KERNELS(CTAMD_ENTRYP)
    CTAMD_ABL_ENTRY(96, 96, 32, 2, 2, 1, LAY_K, LAY_F, 2, 3, 1)
    CTAMD_ABL_ENTRY(96, 96, 32, 2, 2, 1, LAY_K, LAY_F, 2, 3, 2)
    CTAMD_ABL_ENTRY(128, 128, 32, 2, 2, 1, LAY_F, LAY_F, 2, 2, 1)
    CTAMD_ABL_ENTRY(128, 128, 32, 2, 2, 1, LAY_F, LAY_F, 2, 2, 2)};

const GettKernelInfo* gett_f32_kernels(int* count) {
    // register-staged kernels of this file followed by the streaming kernels (stable indices)
    static const std::vector<GettKernelInfo> merged = [] {
        std::vector<GettKernelInfo> v(g_gett_f32_table, g_gett_f32_table + sizeof(g_gett_f32_table) / sizeof(g_gett_f32_table[0]));
        int n = 0;
        const GettKernelInfo* st = gett_f32_stream_kernels(&n);
        v.insert(v.end(), st, st + n);
        return v;
    }();
    *count = (int)merged.size();
    return merged.data();
}

hipError_t launch_splitk_reduce(const SplitKReduceParams& p, hipStream_t stream) {
    const uint32_t Ltot = p.gL.total;
    const size_t total = (size_t)p.gM.total * p.gN.total * Ltot;
    const size_t blocks = (total + 31) / 32;
    if (blocks == 0) return hipSuccess;
    // few slices over many outputs: one lane per four outputs (the partial buffer is 16-byte aligned: a workspace sub-allocation)
    if ((p.gN.total % 4u) == 0u && (p.splitK <= 16u || total >= (1u << 20)) && total >= 4096 &&
        (reinterpret_cast<uintptr_t>(p.partial) & 15u) == 0u) {
        hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, p);
    return hipGetLastError();
}

}  // namespace ctamd
