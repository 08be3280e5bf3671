// reduce.hip — HBM-bound tensor reduction kernels for gfx950.
//
// Replaces the closed kernels behind cutensorReduce (reference call sites:
// cuTENSOR/reduction.cu:219-222 "C_{m,v} = alpha * sum_{h,k} A_{m,h,k,v} + beta * C_{m,v}" :49-61,
// and cuTENSOR/einsum.cu:369-372).
//
// The planner splits A's modes into kept modes (they appear in D) and reduced modes, each a
// mixed-radix group (params.h).  Roofline: HBM; algorithmic bytes = |A| + |D| (+ |C| iff beta != 0),
// reduction.cu:229-231.
//
//   RED_COL      A's stride-1 mode is kept.  One lane owns 4 consecutive kept elements (float4),
//                walks its share of the reduced index space and keeps 4 running values.  Wave
//                loads are 1 KiB contiguous.
//   RED_ROW      A's stride-1 mode is reduced.  One wave owns one kept element, its lanes stride
//                over the reduced space with float4 loads and combine through DPP shuffles.
//   RED_GENERIC  any strides / dtype: one lane per kept element, scalar gathers.
//
// When the kept space alone cannot fill 256 CUs the reduced range is split across workgroups
// (splitR) into a [splitR][kept] partial buffer in the caller's workspace and folded by
// reduce_finalize_kernel, which also applies alpha / beta.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "launch.h"
#include "params.h"
#include "wide_elem.h"

namespace ctamd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { OP_ADD = 3, OP_MUL = 5, OP_MAX = 6, OP_MIN = 7 };   // cutensorOperator_t values

template <typename S> __device__ __forceinline__ S red_identity(int op) {
    switch (op) {
        case OP_MUL: return (S)1;
        case OP_MAX: return -(S)INFINITY;
        case OP_MIN: return (S)INFINITY;
        default: return (S)0;
    }
}
template <typename S> __device__ __forceinline__ S red_apply(int op, S a, S b) {
    switch (op) {
        case OP_MUL: return a * b;
        case OP_MAX: return a > b ? a : b;
        case OP_MIN: return a < b ? a : b;
        default: return a + b;
    }
}

__device__ __forceinline__ uint32_t rd_fast_div(uint32_t n, const FastDiv& d) {
    return __umulhi(n, d.magic) >> d.shift;
}
template <int SLOT>
__device__ __forceinline__ int64_t rd_offset(const ModeGroup& g, uint32_t idx) {
    // fixed trip count, branch-free: padding modes are {d = 1, magic = 0, stride = 0}
    int64_t off = 0;
#pragma unroll
    for (int i = 0; i < kMaxGroupModes; ++i) {
        const uint32_t q = rd_fast_div(idx, g.div[i]);
        off += (int64_t)(idx - q * g.div[i].d) * g.stride[SLOT][i];
        idx = q;
    }
    return off;
}

// ---------------------------------------------------------------------------------------------
// RED_COL, fp32.  grid.x covers kept float4 units, grid.y = splitR.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reduce_col_f32_kernel(const ReduceParams p) {
    const uint32_t unit = blockIdx.x * 256u + threadIdx.x;
    const uint32_t kv = unit * 4u;
    if (kv >= p.kept.total) return;
    const uint32_t split = blockIdx.y;
    const uint32_t rBegin = split * p.redPerSplit;
    uint32_t rEnd = rBegin + p.redPerSplit;
    if (rEnd > p.red.total) rEnd = p.red.total;
    const int op = p.op;
    const float* A = static_cast<const float*>(p.A) + rd_offset<0>(p.kept, kv);
    f32x4 acc;
    for (int e = 0; e < 4; ++e) acc[e] = red_identity<float>(op);
    uint32_t r = rBegin;
    // eight rows (8 x 16 B per lane) in flight per iteration while the reduced index walks ONE mode with a constant stride (the
    // common case: no per-row offset arithmetic between the loads); four rows otherwise
    if (p.red.n == 1) {
        const int64_t step = p.red.stride[0][0];
        const float* q = A + (int64_t)r * step;
        for (; r + 8 <= rEnd; r += 8, q += 8 * step) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q + (int64_t)u * step));
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = red_apply<float>(op, acc[e], v[u][e]);
        }
    }
    for (; r + 4 <= rEnd; r += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A + rd_offset<0>(p.red, r + u)));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = red_apply<float>(op, acc[e], v[u][e]);
    }
    for (; r < rEnd; ++r) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(A + rd_offset<0>(p.red, r));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = red_apply<float>(op, acc[e], v[e]);
    }
    if (p.partial != nullptr) {
        float* P = static_cast<float*>(p.partial) + (size_t)split * p.kept.total + kv;
        *reinterpret_cast<f32x4*>(P) = acc;
        return;
    }
    float*       D = static_cast<float*>(p.D);
    const float* C = static_cast<const float*>(p.C);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // kept mode 0 is contiguous in A; in D / C it may have any stride
        const int64_t oD = rd_offset<1>(p.kept, kv + e);
        float val = p.alpha * acc[e];
        if (p.beta != 0.f) val += p.beta * C[rd_offset<2>(p.kept, kv + e)];
        D[oD] = val;
    }
}

// ---------------------------------------------------------------------------------------------
// RED_ROW, fp32.  One wave per (kept element, split); grid.x covers kept/4 (4 waves per block),
// grid.y = splitR.  redPerSplit is a multiple of 4 and red mode 0 is contiguous with extent % 4 == 0.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reduce_row_f32_kernel(const ReduceParams p) {
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= p.kept.total) return;
    const int lane = threadIdx.x & 63;
    const uint32_t split = blockIdx.y;
    const uint32_t rBegin = split * p.redPerSplit;
    uint32_t rEnd = rBegin + p.redPerSplit;
    if (rEnd > p.red.total) rEnd = p.red.total;
    const int op = p.op;
    const float* A = static_cast<const float*>(p.A) + rd_offset<0>(p.kept, k);
    float acc = red_identity<float>(op);
    uint32_t r = rBegin + 4u * lane;
    for (; r + 3u * 256u < rEnd; r += 4u * 256u) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A + rd_offset<0>(p.red, r + 256u * u)));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = red_apply<float>(op, acc, v[u][e]);
    }
    for (; r < rEnd; r += 256u) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(A + rd_offset<0>(p.red, r));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = red_apply<float>(op, acc, v[e]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc = red_apply<float>(op, acc, __shfl_down(acc, off, 64));
    if (lane != 0) return;
    if (p.partial != nullptr) {
        static_cast<float*>(p.partial)[(size_t)split * p.kept.total + k] = acc;
        return;
    }
    float val = p.alpha * acc;
    if (p.beta != 0.f) val += p.beta * static_cast<const float*>(p.C)[rd_offset<2>(p.kept, k)];
    static_cast<float*>(p.D)[rd_offset<1>(p.kept, k)] = val;
}

// ---------------------------------------------------------------------------------------------
// RED_COL / RED_ROW for every other element type (round 6; wide_elem.h): fp64, complex64, complex128 in the data's own precision,
// bf16 / fp16 with fp32 accumulation.  The fp32 kernels above, with a 16-byte lane = Tr::NV elements (2 / 2 / 1 / 8) and the
// arithmetic of the traits class: ADD / MUL / MAX / MIN on real data, ADD / MUL with conjugation of A on complex data.  Partials are
// [splitR][kept] values of the accumulator type — what reduce_finalize_kernel / reduce_finalize_cplx_kernel fold.
// ---------------------------------------------------------------------------------------------
template <class Tr>
__device__ __forceinline__ void w_finish(const ReduceParams& p, uint32_t k, typename Tr::Acc acc) {
    typedef typename Tr::Elem Elem;
    typename Tr::Acc val = Tr::scale(p.alpha64, p.alphaIm, acc);
    if (p.beta64 != 0.0 || (Tr::CX && p.betaIm != 0.0))
        val = Tr::apply(W_OP_ADD, val, Tr::scale(p.beta64, p.betaIm, Tr::load1(static_cast<const Elem*>(p.C) + rd_offset<2>(p.kept, k), Tr::CX && p.conjC != 0)));
    Tr::store1(static_cast<Elem*>(p.D) + rd_offset<1>(p.kept, k), val);
}

template <class Tr>
__global__ void __launch_bounds__(256) reduce_col_wide_kernel(const ReduceParams p) {
    typedef typename Tr::Elem Elem;
    typedef typename Tr::Acc Acc;
    constexpr int NV = Tr::NV;
    const uint32_t unit = blockIdx.x * 256u + threadIdx.x;
    const uint32_t kv = unit * (uint32_t)NV;
    if (kv >= p.kept.total) return;
    const uint32_t split = blockIdx.y;
    const uint32_t rBegin = split * p.redPerSplit;
    uint32_t rEnd = rBegin + p.redPerSplit;
    if (rEnd > p.red.total) rEnd = p.red.total;
    const int op = p.op;
    const bool conj = Tr::CX && p.conjA != 0;
    const Elem* A = static_cast<const Elem*>(p.A) + rd_offset<0>(p.kept, kv);
    Acc acc[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) acc[e] = Tr::identity(op);
    uint32_t r = rBegin;
    if (p.red.n == 1) {        // one reduced mode: eight rows in flight per lane, addresses one addition apart
        const int64_t step = p.red.stride[0][0];
        const Elem* q = A + (int64_t)r * step;
        for (; r + 8 <= rEnd; r += 8, q += 8 * step) {
            wu32x4 raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) raw[u] = __builtin_nontemporal_load(reinterpret_cast<const wu32x4*>(q + (int64_t)u * step));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                Acc v[NV];
                Tr::unpack(raw[u], v, conj);
#pragma unroll
                for (int e = 0; e < NV; ++e) acc[e] = Tr::apply(op, acc[e], v[e]);
            }
        }
    }
    for (; r + 4 <= rEnd; r += 4) {
        wu32x4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = __builtin_nontemporal_load(reinterpret_cast<const wu32x4*>(A + rd_offset<0>(p.red, r + u)));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Acc v[NV];
            Tr::unpack(raw[u], v, conj);
#pragma unroll
            for (int e = 0; e < NV; ++e) acc[e] = Tr::apply(op, acc[e], v[e]);
        }
    }
    for (; r < rEnd; ++r) {
        Acc v[NV];
        Tr::unpack(*reinterpret_cast<const wu32x4*>(A + rd_offset<0>(p.red, r)), v, conj);
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[e] = Tr::apply(op, acc[e], v[e]);
    }
    if (p.partial != nullptr) {
        Acc* P = static_cast<Acc*>(p.partial) + (size_t)split * p.kept.total + kv;
#pragma unroll
        for (int e = 0; e < NV; ++e) P[e] = acc[e];
        return;
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) w_finish<Tr>(p, kv + e, acc[e]);
}

// one wave per (kept element, split): its lanes stride over the reduced range with 16-byte loads (reduced mode 0 is contiguous, its
// extent and redPerSplit multiples of NV) and meet through lane shuffles
template <class Tr>
__global__ void __launch_bounds__(256) reduce_row_wide_kernel(const ReduceParams p) {
    typedef typename Tr::Elem Elem;
    typedef typename Tr::Acc Acc;
    constexpr int NV = Tr::NV;
    constexpr uint32_t CH = 64u * NV;          // elements one wave-load covers
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= p.kept.total) return;
    const int lane = threadIdx.x & 63;
    const uint32_t split = blockIdx.y;
    const uint32_t rBegin = split * p.redPerSplit;
    uint32_t rEnd = rBegin + p.redPerSplit;
    if (rEnd > p.red.total) rEnd = p.red.total;
    const int op = p.op;
    const bool conj = Tr::CX && p.conjA != 0;
    const Elem* A = static_cast<const Elem*>(p.A) + rd_offset<0>(p.kept, k);
    Acc acc = Tr::identity(op);
    uint32_t r = rBegin + (uint32_t)NV * (uint32_t)lane;
    for (; r + 3u * CH < rEnd; r += 4u * CH) {
        wu32x4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = __builtin_nontemporal_load(reinterpret_cast<const wu32x4*>(A + rd_offset<0>(p.red, r + CH * u)));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Acc v[NV];
            Tr::unpack(raw[u], v, conj);
#pragma unroll
            for (int e = 0; e < NV; ++e) acc = Tr::apply(op, acc, v[e]);
        }
    }
    for (; r < rEnd; r += CH) {
        Acc v[NV];
        Tr::unpack(*reinterpret_cast<const wu32x4*>(A + rd_offset<0>(p.red, r)), v, conj);
#pragma unroll
        for (int e = 0; e < NV; ++e) acc = Tr::apply(op, acc, v[e]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc = Tr::apply(op, acc, Tr::shfl_down(acc, off));
    if (lane != 0) return;
    if (p.partial != nullptr) {
        static_cast<Acc*>(p.partial)[(size_t)split * p.kept.total + k] = acc;
        return;
    }
    w_finish<Tr>(p, k, acc);
}

template <class Tr>
static void launch_wide_t(const ReduceParams& p, int variant, hipStream_t stream) {
    if (variant == RED_COL) {
        const dim3 grid(((p.kept.total / (uint32_t)Tr::NV) + 255u) / 256u, p.splitR);
        hipLaunchKernelGGL(reduce_col_wide_kernel<Tr>, grid, dim3(256), 0, stream, p);
    } else {
        const dim3 grid((p.kept.total + 3u) / 4u, p.splitR);
        hipLaunchKernelGGL(reduce_row_wide_kernel<Tr>, grid, dim3(256), 0, stream, p);
    }
}

// ---------------------------------------------------------------------------------------------
// RED_GENERIC: any dtype / strides.  One lane per (kept element, split).
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ double rg_load(const T* p) { return (double)(*p); }
template <> __device__ __forceinline__ double rg_load<__half>(const __half* p) { return (double)__half2float(*p); }
template <> __device__ __forceinline__ double rg_load<__hip_bfloat16>(const __hip_bfloat16* p) { return (double)__bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void rg_store(T* p, double v) { *p = (T)v; }
template <> __device__ __forceinline__ void rg_store<__half>(__half* p, double v) { *p = __float2half((float)v); }
template <> __device__ __forceinline__ void rg_store<__hip_bfloat16>(__hip_bfloat16* p, double v) { *p = __float2bfloat16((float)v); }

template <typename T, typename S>
__global__ void __launch_bounds__(256) reduce_generic_kernel(const ReduceParams p) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= p.kept.total) return;
    const uint32_t split = blockIdx.y;
    const uint32_t rBegin = split * p.redPerSplit;
    uint32_t rEnd = rBegin + p.redPerSplit;
    if (rEnd > p.red.total) rEnd = p.red.total;
    const int op = p.op;
    const T* A = static_cast<const T*>(p.A) + rd_offset<0>(p.kept, k);
    S acc = red_identity<S>(op);
    // (round 6: eight / four loads in flight per lane — the loop used to issue one load and wait for it, 1.5 TB/s on 'abc->ac' at odd
    // extents where neighbouring lanes DO read neighbouring elements; same order of the combines, same bits)
    uint32_t r = rBegin;
    if (p.red.n == 1) {
        const int64_t step = p.red.stride[0][0];
        const T* q = A + (int64_t)r * step;
        for (; r + 8 <= rEnd; r += 8, q += 8 * step) {
            S v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (S)rg_load<T>(q + (int64_t)u * step);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = red_apply<S>(op, acc, v[u]);
        }
    }
    for (; r + 4 <= rEnd; r += 4) {
        S v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (S)rg_load<T>(A + rd_offset<0>(p.red, r + u));
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = red_apply<S>(op, acc, v[u]);
    }
    for (; r < rEnd; ++r) acc = red_apply<S>(op, acc, (S)rg_load<T>(A + rd_offset<0>(p.red, r)));
    if (p.partial != nullptr) {
        static_cast<S*>(p.partial)[(size_t)split * p.kept.total + k] = acc;
        return;
    }
    const S alpha = sizeof(S) == 8 ? (S)p.alpha64 : (S)p.alpha;
    const S beta  = sizeof(S) == 8 ? (S)p.beta64 : (S)p.beta;
    S val = alpha * acc;
    if (beta != (S)0) val += beta * (S)rg_load<T>(static_cast<const T*>(p.C) + rd_offset<2>(p.kept, k));
    rg_store<T>(static_cast<T*>(p.D) + rd_offset<1>(p.kept, k), (double)val);
}

// RED_GENERIC with A's stride-1 mode REDUCED (ReduceParams::rowAny, round 6): 'abc->bc', 'ab->b' at extents / alignments the 16-byte-lane
// row kernel refuses.  One lane per kept element (above) puts neighbouring lanes a kept stride apart — 0.4-0.6 TB/s; here one WAVE owns a
// kept element, its lanes stride over the reduced range element by element (coalesced), four loads in flight, and meet through lane
// shuffles.  Partials as above ([splitR][kept] accumulators).
template <typename T, typename S>
__global__ void __launch_bounds__(256) reduce_row_any_kernel(const ReduceParams p) {
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (k >= p.kept.total) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t split = blockIdx.y;
    const uint32_t rBegin = split * p.redPerSplit;
    uint32_t rEnd = rBegin + p.redPerSplit;
    if (rEnd > p.red.total) rEnd = p.red.total;
    const int op = p.op;
    const T* A = static_cast<const T*>(p.A) + rd_offset<0>(p.kept, k);
    S acc = red_identity<S>(op);
    uint32_t r = rBegin + lane;
    for (; r + 3u * 64u < rEnd; r += 4u * 64u) {
        S v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (S)rg_load<T>(A + rd_offset<0>(p.red, r + 64u * (uint32_t)u));
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = red_apply<S>(op, acc, v[u]);
    }
    for (; r < rEnd; r += 64u) acc = red_apply<S>(op, acc, (S)rg_load<T>(A + rd_offset<0>(p.red, r)));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc = red_apply<S>(op, acc, __shfl_down(acc, off, 64));
    if (lane != 0u) return;
    if (p.partial != nullptr) {
        static_cast<S*>(p.partial)[(size_t)split * p.kept.total + k] = acc;
        return;
    }
    const S alpha = sizeof(S) == 8 ? (S)p.alpha64 : (S)p.alpha;
    const S beta  = sizeof(S) == 8 ? (S)p.beta64 : (S)p.beta;
    S val = alpha * acc;
    if (beta != (S)0) val += beta * (S)rg_load<T>(static_cast<const T*>(p.C) + rd_offset<2>(p.kept, k));
    rg_store<T>(static_cast<T*>(p.D) + rd_offset<1>(p.kept, k), (double)val);
}

// D[k] = alpha * combine_s partial[s][k] + beta * C[k]
template <typename T, typename S>
__global__ void __launch_bounds__(256) reduce_finalize_kernel(const ReduceParams p) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= p.kept.total) return;
    const int op = p.op;
    const S* P = static_cast<const S*>(p.partial) + k;
    S acc = red_identity<S>(op);
    for (uint32_t s = 0; s < p.splitR; ++s) acc = red_apply<S>(op, acc, P[(size_t)s * p.kept.total]);
    const S alpha = sizeof(S) == 8 ? (S)p.alpha64 : (S)p.alpha;
    const S beta  = sizeof(S) == 8 ? (S)p.beta64 : (S)p.beta;
    S val = alpha * acc;
    if (beta != (S)0) val += beta * (S)rg_load<T>(static_cast<const T*>(p.C) + rd_offset<2>(p.kept, k));
    rg_store<T>(static_cast<T*>(p.D) + rd_offset<1>(p.kept, k), (double)val);
}

// ---------------------------------------------------------------------------------------------
// Complex reductions (HIP_C_32F / HIP_C_64F; python/einsum.h:326-343,430-441 runs a unary einsum on complex tensors through
// cutensorCreateReduction(OP_ADD) + cutensorReduce; einsum.cu:346-372): RED_GENERIC's structure on (re, im) pairs — one lane per
// (kept element, split), complex alpha / beta, conjugation of A / C, ADD and MUL.  Accumulation in the data's own precision
// (like the real kernels: float for complex64, double for complex128); partials are [splitR][kept] pairs.
// ---------------------------------------------------------------------------------------------
template <typename R> struct RdCx { R re, im; };
template <typename R> __device__ __forceinline__ RdCx<R> rc_mul(RdCx<R> a, RdCx<R> b) { return RdCx<R>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <typename R> __device__ __forceinline__ RdCx<R> rc_apply(int op, RdCx<R> a, RdCx<R> b) {
    return op == OP_MUL ? rc_mul(a, b) : RdCx<R>{a.re + b.re, a.im + b.im};
}
template <typename R> __device__ __forceinline__ void rc_finish(const ReduceParams& p, uint32_t k, RdCx<R> acc) {
    const RdCx<R> alpha = {(R)p.alpha64, (R)p.alphaIm}, beta = {(R)p.beta64, (R)p.betaIm};
    RdCx<R> val = rc_mul(alpha, acc);
    if (beta.re != (R)0 || beta.im != (R)0) {
        RdCx<R> c = static_cast<const RdCx<R>*>(p.C)[rd_offset<2>(p.kept, k)];
        if (p.conjC) c.im = -c.im;
        const RdCx<R> bc = rc_mul(beta, c);
        val.re += bc.re; val.im += bc.im;
    }
    static_cast<RdCx<R>*>(p.D)[rd_offset<1>(p.kept, k)] = val;
}

template <typename R>
__global__ void __launch_bounds__(256) reduce_generic_cplx_kernel(const ReduceParams p) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= p.kept.total) return;
    const uint32_t split = blockIdx.y;
    const uint32_t rBegin = split * p.redPerSplit;
    uint32_t rEnd = rBegin + p.redPerSplit;
    if (rEnd > p.red.total) rEnd = p.red.total;
    const int op = p.op;
    const RdCx<R>* A = static_cast<const RdCx<R>*>(p.A) + rd_offset<0>(p.kept, k);
    RdCx<R> acc = {op == OP_MUL ? (R)1 : (R)0, (R)0};
    const R sgn = p.conjA ? (R)-1 : (R)1;
    uint32_t r = rBegin;
    for (; r + 4 <= rEnd; r += 4) {               // four loads in flight (reduce_generic_kernel, round 6); same order of the combines
        RdCx<R> x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = A[rd_offset<0>(p.red, r + (uint32_t)u)];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x[u].im *= sgn; acc = rc_apply<R>(op, acc, x[u]); }
    }
    for (; r < rEnd; ++r) {
        RdCx<R> x = A[rd_offset<0>(p.red, r)];
        x.im *= sgn;
        acc = rc_apply<R>(op, acc, x);
    }
    if (p.partial != nullptr) {
        static_cast<RdCx<R>*>(p.partial)[(size_t)split * p.kept.total + k] = acc;
        return;
    }
    rc_finish<R>(p, k, acc);
}

template <typename R>
__global__ void __launch_bounds__(256) reduce_finalize_cplx_kernel(const ReduceParams p) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= p.kept.total) return;
    const int op = p.op;
    const RdCx<R>* P = static_cast<const RdCx<R>*>(p.partial) + k;
    RdCx<R> acc = {op == OP_MUL ? (R)1 : (R)0, (R)0};
    for (uint32_t s = 0; s < p.splitR; ++s) acc = rc_apply<R>(op, acc, P[(size_t)s * p.kept.total]);
    rc_finish<R>(p, k, acc);
}

template <typename T, typename S>
static void launch_generic_t(const ReduceParams& p, hipStream_t stream) {
    if (p.rowAny != 0u) {                         // A's stride-1 mode is reduced: a wave per kept element (reduce_row_any_kernel)
        const dim3 grid((p.kept.total + 3u) / 4u, p.splitR);
        hipLaunchKernelGGL((reduce_row_any_kernel<T, S>), grid, dim3(256), 0, stream, p);
        return;
    }
    const dim3 grid((p.kept.total + 255u) / 256u, p.splitR);
    hipLaunchKernelGGL((reduce_generic_kernel<T, S>), grid, dim3(256), 0, stream, p);
}
template <typename T, typename S>
static void launch_finalize_t(const ReduceParams& p, hipStream_t stream) {
    hipLaunchKernelGGL((reduce_finalize_kernel<T, S>), dim3((p.kept.total + 255u) / 256u), dim3(256), 0, stream, p);
}

hipError_t launch_reduce(const ReduceParams& p, int variant, int dtype, bool acc64, hipStream_t stream) {
    if (p.kept.total == 0) return hipSuccess;
    if (variant == RED_COL && dtype == HIP_R_32F) {
        const dim3 grid(((p.kept.total / 4u) + 255u) / 256u, p.splitR);
        hipLaunchKernelGGL(reduce_col_f32_kernel, grid, dim3(256), 0, stream, p);
    } else if (variant == RED_ROW && dtype == HIP_R_32F) {
        const dim3 grid((p.kept.total + 3u) / 4u, p.splitR);
        hipLaunchKernelGGL(reduce_row_f32_kernel, grid, dim3(256), 0, stream, p);
    } else if ((variant == RED_COL || variant == RED_ROW) && dtype != HIP_R_32F && (!acc64 || dtype == HIP_R_64F || dtype == HIP_C_64F)) {
        switch (dtype) {       // the tiled kernels of the other element types (wide_elem.h)
            case HIP_R_64F:  launch_wide_t<WF64>(p, variant, stream); break;
            case HIP_R_16F:  launch_wide_t<WH16<false>>(p, variant, stream); break;
            case HIP_R_16BF: launch_wide_t<WH16<true>>(p, variant, stream); break;
            case HIP_C_32F:  launch_wide_t<WCplx<float>>(p, variant, stream); break;
            case HIP_C_64F:  launch_wide_t<WCplx<double>>(p, variant, stream); break;
            default: return hipErrorInvalidValue;
        }
    } else if (variant == RED_GENERIC) {
        switch (dtype) {
            case HIP_R_32F:  if (acc64) launch_generic_t<float, double>(p, stream); else launch_generic_t<float, float>(p, stream); break;
            case HIP_R_64F:  launch_generic_t<double, double>(p, stream); break;
            case HIP_R_16F:  launch_generic_t<__half, float>(p, stream); break;
            case HIP_R_16BF: launch_generic_t<__hip_bfloat16, float>(p, stream); break;
            case HIP_C_32F:  hipLaunchKernelGGL(reduce_generic_cplx_kernel<float>, dim3((p.kept.total + 255u) / 256u, p.splitR), dim3(256), 0, stream, p); break;
            case HIP_C_64F:  hipLaunchKernelGGL(reduce_generic_cplx_kernel<double>, dim3((p.kept.total + 255u) / 256u, p.splitR), dim3(256), 0, stream, p); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_reduce_finalize(const ReduceParams& p, int dtype, bool acc64, hipStream_t stream) {
    if (p.kept.total == 0) return hipSuccess;
    switch (dtype) {
        case HIP_R_32F:  if (acc64) launch_finalize_t<float, double>(p, stream); else launch_finalize_t<float, float>(p, stream); break;
        case HIP_R_64F:  launch_finalize_t<double, double>(p, stream); break;
        case HIP_R_16F:  launch_finalize_t<__half, float>(p, stream); break;
        case HIP_R_16BF: launch_finalize_t<__hip_bfloat16, float>(p, stream); break;
        case HIP_C_32F:  hipLaunchKernelGGL(reduce_finalize_cplx_kernel<float>, dim3((p.kept.total + 255u) / 256u), dim3(256), 0, stream, p); break;
        case HIP_C_64F:  hipLaunchKernelGGL(reduce_finalize_cplx_kernel<double>, dim3((p.kept.total + 255u) / 256u), dim3(256), 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ctamd
