// gett_simple.hip — portable-precision contraction kernel for every dtype the MFMA paths do not
// cover yet (fp64, fp16, bf16 with fp32 accumulation) — same GEMM view and mode-group addressing
// as gett_f32.hip, classic 16 x 16 LDS tile, one output element per lane, VALU FMAs.
//
// It exists so that the ABI accepts the type matrix the reference exercises
// (cuTENSOR/einsum.cu:36-55: double / float / __half; cuTENSOR/python/cutensor/torch/einsum.cc:28-54
// adds bf16) with correct results; it is not a performance path.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "launch.h"
#include "params.h"

namespace ctamd {

__device__ __forceinline__ uint32_t gs_fast_div(uint32_t n, const FastDiv& d) {
    return __umulhi(n, d.magic) >> d.shift;
}
template <int SLOT>
__device__ __forceinline__ int64_t gs_offset(const ModeGroup& g, uint32_t idx) {
    int64_t off = 0;
#pragma unroll
    for (int i = 0; i < kMaxGroupModes; ++i) {
        const uint32_t q = gs_fast_div(idx, g.div[i]);
        off += (int64_t)(idx - q * g.div[i].d) * g.stride[SLOT][i];
        idx = q;
    }
    return off;
}
__device__ __forceinline__ int64_t gs_offset_c(const ModeGroup& g, const int64_t* cs, uint32_t idx) {
    int64_t off = 0;
#pragma unroll
    for (int i = 0; i < kMaxGroupModes; ++i) {
        const uint32_t q = gs_fast_div(idx, g.div[i]);
        off += (int64_t)(idx - q * g.div[i].d) * cs[i];
        idx = q;
    }
    return off;
}

template <typename T> __device__ __forceinline__ double gs_load(const T* p) { return (double)(*p); }
template <> __device__ __forceinline__ double gs_load<__half>(const __half* p) { return (double)__half2float(*p); }
template <> __device__ __forceinline__ double gs_load<__hip_bfloat16>(const __hip_bfloat16* p) { return (double)__bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void gs_store(T* p, double v) { *p = (T)v; }
template <> __device__ __forceinline__ void gs_store<__half>(__half* p, double v) { *p = __float2half((float)v); }
template <> __device__ __forceinline__ void gs_store<__hip_bfloat16>(__hip_bfloat16* p, double v) { *p = __float2bfloat16((float)v); }

template <typename T, typename S>
__global__ void __launch_bounds__(256) gett_simple_kernel(const GettParams p) {
    __shared__ S As[16][17];
    __shared__ S Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    uint32_t id = blockIdx.x;
    const uint32_t mt = id % p.tilesM; id /= p.tilesM;
    const uint32_t nt = id % p.tilesN;
    const uint32_t l = id / p.tilesN;
    const T* A = static_cast<const T*>(p.A);
    const T* B = static_cast<const T*>(p.B);
    const T* C = static_cast<const T*>(p.C);
    T*       D = static_cast<T*>(p.D);
    A += gs_offset<0>(p.gL, l);
    B += gs_offset<1>(p.gL, l);
    D += gs_offset<2>(p.gL, l);
    C += gs_offset_c(p.gL, p.cStrideL, l);
    const uint32_t m = mt * 16 + ty, n = nt * 16 + tx;
    const bool okM = m < p.gM.total, okN = n < p.gN.total;
    const int64_t offAm = okM ? gs_offset<0>(p.gM, m) : 0;
    const int64_t offBn = okN ? gs_offset<0>(p.gN, n) : 0;
    const uint32_t K = p.gK.total;
    S acc = (S)0;
    for (uint32_t kt = 0; kt < K; kt += 16) {
        // As[ty][tx] = A[m0 + ty][kt + tx];  Bs[ty][tx] = B[kt + ty][n0 + tx]
        const uint32_t ka = kt + tx, kb = kt + ty;
        S a = (S)0, b = (S)0;
        if (okM && ka < K) a = (S)gs_load<T>(A + offAm + gs_offset<0>(p.gK, ka));
        if (okN && kb < K) b = (S)gs_load<T>(B + offBn + gs_offset<1>(p.gK, kb));
        As[ty][tx] = a;
        Bs[ty][tx] = b;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc += As[ty][kk] * Bs[kk][tx];
        __syncthreads();
    }
    if (!okM || !okN) return;
    const S alpha = sizeof(S) == 8 ? (S)p.alpha64 : (S)p.alpha;
    const S beta  = sizeof(S) == 8 ? (S)p.beta64 : (S)p.beta;
    S val = alpha * acc;
    if (beta != (S)0)
        val += beta * (S)gs_load<T>(C + gs_offset_c(p.gM, p.cStrideM, m) + gs_offset_c(p.gN, p.cStrideN, n));
    gs_store<T>(D + gs_offset<1>(p.gM, m) + gs_offset<1>(p.gN, n), (double)val);
}

// ---------------------------------------------------------------------------------------------
// Wide-mode contraction: one output element per lane, the mode table is read through wave-uniform indices
// (scalar loads).  Output index -> digits once per element (64-bit arithmetic), contracted index -> digits per
// k with multiply-high divisions.  Correctness path for tensors the tiled kernels cannot describe.
// ---------------------------------------------------------------------------------------------
template <typename T, typename S>
__global__ void __launch_bounds__(256) gett_wide_kernel(const WideParams p) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= p.outTotal) return;
    int64_t oA = 0, oB = 0, oC = 0, oD = 0;
    uint64_t rem = e;
    for (uint32_t i = 0; i < p.nOut; ++i) {
        const WideMode m = p.modes[i];
        const uint64_t q = rem / m.div.d;
        const int64_t digit = (int64_t)(rem - q * m.div.d);
        oA += digit * m.sA; oB += digit * m.sB; oC += digit * m.sC; oD += digit * m.sD;
        rem = q;
    }
    const T* A = static_cast<const T*>(p.A) + oA;
    const T* B = static_cast<const T*>(p.B) + oB;
    S acc = (S)0;
    for (uint32_t k = 0; k < p.kTotal; ++k) {
        int64_t ka = 0, kb = 0;
        uint32_t r = k;
        for (uint32_t i = 0; i < p.nK; ++i) {
            const WideMode m = p.modes[p.nOut + i];
            const uint32_t q = (m.div.d < 2) ? r : gs_fast_div(r, m.div);
            const int64_t digit = (int64_t)(r - q * m.div.d);
            ka += digit * m.sA; kb += digit * m.sB;
            r = q;
        }
        acc += (S)gs_load<T>(A + ka) * (S)gs_load<T>(B + kb);
    }
    const S alpha = sizeof(S) == 8 ? (S)p.alpha64 : (S)p.alpha;
    const S beta  = sizeof(S) == 8 ? (S)p.beta64 : (S)p.beta;
    S val = alpha * acc;
    if (beta != (S)0) val += beta * (S)gs_load<T>(static_cast<const T*>(p.C) + oC);
    gs_store<T>(static_cast<T*>(p.D) + oD, (double)val);
}

// Complex data: D = alpha * op(A) * op(B) + beta * op(C) with op = identity or conjugate
// (cuTENSOR/contraction_jit.cu:31-41: std::complex<float> tensors and scalars).  R = float or double.
template <typename R>
__global__ void __launch_bounds__(256) gett_wide_complex_kernel(const WideParams p) {
    struct Cx { R re, im; };
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= p.outTotal) return;
    int64_t oA = 0, oB = 0, oC = 0, oD = 0;
    uint64_t rem = e;
    for (uint32_t i = 0; i < p.nOut; ++i) {
        const WideMode m = p.modes[i];
        const uint64_t q = rem / m.div.d;
        const int64_t digit = (int64_t)(rem - q * m.div.d);
        oA += digit * m.sA; oB += digit * m.sB; oC += digit * m.sC; oD += digit * m.sD;
        rem = q;
    }
    const Cx* A = static_cast<const Cx*>(p.A) + oA;
    const Cx* B = static_cast<const Cx*>(p.B) + oB;
    const R sa = p.conjA ? (R)-1 : (R)1, sb = p.conjB ? (R)-1 : (R)1;
    R accRe = 0, accIm = 0;
    for (uint32_t k = 0; k < p.kTotal; ++k) {
        int64_t ka = 0, kb = 0;
        uint32_t r = k;
        for (uint32_t i = 0; i < p.nK; ++i) {
            const WideMode m = p.modes[p.nOut + i];
            const uint32_t q = (m.div.d < 2) ? r : gs_fast_div(r, m.div);
            const int64_t digit = (int64_t)(r - q * m.div.d);
            ka += digit * m.sA; kb += digit * m.sB;
            r = q;
        }
        const Cx a = A[ka], b = B[kb];
        const R aim = sa * a.im, bim = sb * b.im;
        accRe += a.re * b.re - aim * bim;
        accIm += a.re * bim + aim * b.re;
    }
    const R alRe = (R)p.alpha64, alIm = (R)p.alphaIm, beRe = (R)p.beta64, beIm = (R)p.betaIm;
    Cx out;
    out.re = alRe * accRe - alIm * accIm;
    out.im = alRe * accIm + alIm * accRe;
    if (beRe != (R)0 || beIm != (R)0) {
        Cx c = static_cast<const Cx*>(p.C)[oC];
        if (p.conjC) c.im = -c.im;
        out.re += beRe * c.re - beIm * c.im;
        out.im += beRe * c.im + beIm * c.re;
    }
    static_cast<Cx*>(p.D)[oD] = out;
}

template <typename T, typename S>
static void launch_wide_t(const WideParams& p, hipStream_t stream) {
    const uint64_t blocks = (p.outTotal + 255) / 256;
    hipLaunchKernelGGL((gett_wide_kernel<T, S>), dim3((unsigned)blocks), dim3(256), 0, stream, p);
}

hipError_t launch_gett_wide(const WideParams& p, int dtype, bool accumulate64, hipStream_t stream) {
    if (p.outTotal == 0) return hipSuccess;
    if ((p.outTotal + 255) / 256 >= (1ull << 31)) return hipErrorInvalidValue;
    switch (dtype) {
        case HIP_R_32F:  if (accumulate64) launch_wide_t<float, double>(p, stream); else launch_wide_t<float, float>(p, stream); break;
        case HIP_R_64F:  launch_wide_t<double, double>(p, stream); break;
        case HIP_R_16F:  launch_wide_t<__half, float>(p, stream); break;
        case HIP_R_16BF: launch_wide_t<__hip_bfloat16, float>(p, stream); break;
        case HIP_C_32F:
            hipLaunchKernelGGL(gett_wide_complex_kernel<float>, dim3((unsigned)((p.outTotal + 255) / 256)), dim3(256), 0, stream, p);
            break;
        case HIP_C_64F:
            hipLaunchKernelGGL(gett_wide_complex_kernel<double>, dim3((unsigned)((p.outTotal + 255) / 256)), dim3(256), 0, stream, p);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <typename T, typename S>
static void launch_simple_t(const GettParams& p, hipStream_t stream) {
    hipLaunchKernelGGL((gett_simple_kernel<T, S>), dim3(p.nBlocks), dim3(256), 0, stream, p);
}

hipError_t launch_gett_simple(const GettParams& p, int dtype, bool accumulate64, hipStream_t stream) {
    if (p.nBlocks == 0) return hipSuccess;
    switch (dtype) {
        case HIP_R_32F:  if (accumulate64) launch_simple_t<float, double>(p, stream); else launch_simple_t<float, float>(p, stream); break;
        case HIP_R_64F:  launch_simple_t<double, double>(p, stream); break;
        case HIP_R_16F:  launch_simple_t<__half, float>(p, stream); break;
        case HIP_R_16BF: launch_simple_t<__hip_bfloat16, float>(p, stream); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ctamd
