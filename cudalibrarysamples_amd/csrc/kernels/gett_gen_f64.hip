// gett_gen_f64.hip — fp64 instantiations of the general MFMA GETT kernel (gett_gen.inc): v_mfma_f64_16x16x4_f64.
//   V = 2: 16-byte lanes (two doubles) — 128 x 128 x 16 and 64 x 64 x 16 tiles;  V = 1: 8-byte gathers — 64 x 64 x 16
#include "gett_gen.inc"

namespace ctamd {

static const GettKernelInfo g_gen_f64_table[] = {
    CTAMD_GEN_ORIENTS(GEN_F64, 128, 128, 16, 2)
    CTAMD_GEN_ORIENTS(GEN_F64, 64, 64, 16, 2)
    CTAMD_GEN_ORIENTS(GEN_F64, 64, 64, 16, 1)};

const GettKernelInfo* gett_gen_f64_kernels(int* count) {
    *count = (int)(sizeof(g_gen_f64_table) / sizeof(g_gen_f64_table[0]));
    return g_gen_f64_table;
}

}  // namespace ctamd

// ---------------------------------------------------------------------------------------------
// Split-K fold of the fp64 / complex general kernels: one lane per output element (n fastest: the partial reads and, when D's
// stride-1 mode is kernel-N, the stores are coalesced), slices summed in sequence, D = alpha * sum + beta * op(C) in the
// accumulator precision.  R = float (complex64) or double; CPLX: elements are (re, im) pairs.
// ---------------------------------------------------------------------------------------------
namespace ctamd {
template <typename R, bool CPLX>
__global__ void __launch_bounds__(256) gen_splitk_reduce_kernel(const SplitKReduceParams p) {
    constexpr int W = CPLX ? 2 : 1;
    const uint32_t Mtot = p.gM.total, Ntot = p.gN.total;
    const size_t plane = (size_t)Mtot * Ntot, total = plane * p.gL.total;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const R* src = reinterpret_cast<const R*>(p.partial) + e * W;
    R re = 0, im = 0;
    for (uint32_t s = 0; s < p.splitK; ++s) {
        re += src[(size_t)s * total * W];
        if constexpr (CPLX) im += src[(size_t)s * total * W + 1];
    }
    const uint32_t l = (uint32_t)(e / plane);
    const size_t rem = e - (size_t)l * plane;
    const uint32_t m = (uint32_t)(rem / Ntot), n = (uint32_t)(rem - (size_t)m * Ntot);
    int64_t oDl, oCl, oDm, oCm, oDn, oCn;
    group_offset2<2>(p.gL, p.cStrideL, l, oDl, oCl);
    group_offset2<1>(p.gM, p.cStrideM, m, oDm, oCm);
    group_offset2<1>(p.gN, p.cStrideN, n, oDn, oCn);
    const int64_t oD = oDl + oDm + oDn, oC = oCl + oCm + oCn;
    const R alRe = (R)p.alpha64, alIm = (R)p.alphaIm, beRe = (R)p.beta64, beIm = (R)p.betaIm;
    if constexpr (!CPLX) {
        R val = alRe * re;
        if (beRe != (R)0) val += beRe * static_cast<const R*>(p.C)[oC];
        static_cast<R*>(p.D)[oD] = val;
    } else {
        R oRe = alRe * re - alIm * im, oIm = alRe * im + alIm * re;
        if (beRe != (R)0 || beIm != (R)0) {
            const R* c = static_cast<const R*>(p.C) + 2 * oC;
            const R cRe = c[0], cIm = p.conjC ? -c[1] : c[1];
            oRe += beRe * cRe - beIm * cIm;
            oIm += beRe * cIm + beIm * cRe;
        }
        R* d = static_cast<R*>(p.D) + 2 * oD;
        d[0] = oRe;
        d[1] = oIm;
    }
}

hipError_t launch_gen_splitk_reduce(const SplitKReduceParams& p, int elem, hipStream_t stream) {
    const size_t total = (size_t)p.gM.total * p.gN.total * p.gL.total;
    if (total == 0) return hipSuccess;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    switch (elem) {
        case GEN_F64: hipLaunchKernelGGL((gen_splitk_reduce_kernel<double, false>), grid, block, 0, stream, p); break;
        case GEN_C32: hipLaunchKernelGGL((gen_splitk_reduce_kernel<float, true>), grid, block, 0, stream, p); break;
        case GEN_C64: hipLaunchKernelGGL((gen_splitk_reduce_kernel<double, true>), grid, block, 0, stream, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
}  // namespace ctamd

// ---------------------------------------------------------------------------------------------
// What the chip sustains on nothing but v_mfma_f64_16x16x4_f64 (the fp64 general kernels' roofline; the microarchitecture guide
// quotes no fp64 matrix figure, so it is measured): 256 threads per CU, eight independent accumulators per wave, operands held
// in registers.  dataKind 0: zeros (issue rate at full clock), 1: U(-1, 1).  Returns TFLOP/s (2 * 16 * 16 * 4 flop per MFMA).
// ---------------------------------------------------------------------------------------------
namespace ctamd {
__global__ void __launch_bounds__(256) mfma_f64_ceiling_kernel(const double* __restrict__ src, double* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[tid + 256 * i]; b[i] = src[tid + 256 * (4 + i)]; }
    f64x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f64x4{0., 0., 0., 0.};
    // inline asm: through the builtin hipcc keeps the accumulators in AGPRs and copies all 64 registers to and from VGPRs in every
    // iteration (128 moves per 8 MFMAs: the first form of this kernel measured 36 TFLOP/s where the GETT kernel itself reaches 62).
    // Eight independent accumulators, each touched once per 8 MFMAs: no hazard to pad.
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 1) & 3]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results land before anything reads them
    f64x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[(size_t)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}
}  // namespace ctamd

extern "C" int ctamdMeasureMfmaCeilingF64(int dataKind, float* tflops) {
    using namespace ctamd;
    if (tflops == nullptr) return -1;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    const int cus = prop.multiProcessorCount;
    const size_t n = 256 * 8;
    double h[256 * 8];
    uint32_t lcg = 12345u;
    for (size_t i = 0; i < n; ++i) {
        lcg = lcg * 1664525u + 1013904223u;
        h[i] = dataKind == 0 ? 0.0 : 2.0 * (double)(lcg >> 8) / 16777216.0 - 1.0;
    }
    double *d = nullptr, *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = -1;
    if (hipMalloc((void**)&d, n * 8) == hipSuccess && hipMalloc((void**)&out, (size_t)cus * 256 * 8) == hipSuccess &&
        hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        const int iters = 5000;
        auto launch = [&]() { hipLaunchKernelGGL(mfma_f64_ceiling_kernel, dim3(cus), dim3(256), 0, nullptr, d, out, iters); };
        for (int w = 0; w < 3; ++w) launch();
        (void)hipEventRecord(e0, nullptr);
        for (int w = 0; w < 3; ++w) launch();
        (void)hipEventRecord(e1, nullptr);
        float ms = 0.f;
        if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f) {
            const double flops = 3.0 * (double)cus * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
            *tflops = (float)(flops / (ms * 1e-3) / 1e12);
            rc = 0;
        }
    }
    (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d) (void)hipFree(d);
    if (out) (void)hipFree(out);
    return rc;
}
