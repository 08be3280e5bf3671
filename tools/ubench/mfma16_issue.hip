// Microbenchmark (tools only): at what interval can ONE wave per SIMD issue independent v_mfma_f32_16x16x32_bf16 (4 passes =
// 16 cycles on paper)?  tools/ubench/mfma_shape_power.hip measured 56 % of the nominal rate for a plain stream of them
// ("issue-limited"), yet the vendor's best bf16 GEMM kernel runs on this shape at 90 %.  Variants of the stream, zero operands
// (no power limit), one workgroup of 4 waves per CU; the clock is taken from a v_mfma_f32_32x32x16_bf16 stream (32 cycles each):
//   0  32x32x16, 8 accumulators in AGPRs                                  (reference)
//   1  16x16x32, 16 accumulators, compiler's choice of registers (builtin)
//   2  16x16x32, 16 accumulators forced into AGPRs (inline asm, "+a")
//   3  16x16x32, 16 accumulators forced into VGPRs ("+v")
//   4  as 2 with 32 accumulators
//   5  as 2, two different A / B register quads alternating
//   6  as 2 with an s_nop 0 behind every MFMA
//   7  as 2 with a VALU v_mov behind every second MFMA
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma16_issue.hip -o tools/ubench/mfma16_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256, 1) probe(const s16x8* __restrict__ data, float* out, int iters) {
    s16x8 a0 = data[threadIdx.x], b0 = data[256 + threadIdx.x], a1 = data[512 + threadIdx.x], b1 = data[768 + threadIdx.x];
    float r = 0.f;
    if constexpr (MODE == 0) {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[i & 7], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][7];
    } else {
        constexpr int NACC = (MODE == 4) ? 32 : 16;
        f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {   // 32 x (16x16x32) = the flops of 16 x (32x32x16)
                if constexpr (MODE == 1) {
                    acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[i % NACC], 0, 0, 0);
                } else if constexpr (MODE == 3) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i % NACC]) : "v"(a0), "v"(b0));
                } else if constexpr (MODE == 5) {
                    if (i & 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i % NACC]) : "v"(a1), "v"(b1));
                    else       asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i % NACC]) : "v"(a0), "v"(b0));
                } else if constexpr (MODE == 6) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 0" : "+a"(acc[i % NACC]) : "v"(a0), "v"(b0));
                } else if constexpr (MODE == 7) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i % NACC]) : "v"(a0), "v"(b0));
                    if (i & 1) asm volatile("v_mov_b32 %0, %0" : "+v"(r));
                } else {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i % NACC]) : "v"(a0), "v"(b0));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
static double run_ms(const s16x8* d, float* out, int cus, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) probe<MODE><<<cus, 256>>>(d, out, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) probe<MODE><<<cus, 256>>>(d, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / 3.0;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    s16x8* d; float* out;
    hipMalloc(&d, 1024 * sizeof(s16x8)); hipMemset(d, 0, 1024 * sizeof(s16x8));
    hipMalloc(&out, (size_t)cus * 256 * 4);
    const int iters = 20000;
    const double flops = (double)cus * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    const double t0 = run_ms<0>(d, out, cus, iters);
    const double mhz = iters * 16 * 32.0 / (t0 * 1e-3) / 1e6;    // 32 cycles per 32x32x16
    printf("{\"mode\": 0, \"ms\": %.3f, \"tflops\": %.0f, \"clock_mhz_if_32_cycles_each\": %.0f}\n", t0, flops / (t0 * 1e-3) / 1e12, mhz);
    const double t[7] = {run_ms<1>(d, out, cus, iters), run_ms<2>(d, out, cus, iters), run_ms<3>(d, out, cus, iters), run_ms<4>(d, out, cus, iters),
                         run_ms<5>(d, out, cus, iters), run_ms<6>(d, out, cus, iters), run_ms<7>(d, out, cus, iters)};
    for (int m = 0; m < 7; ++m)
        printf("{\"mode\": %d, \"ms\": %.3f, \"tflops\": %.0f, \"cycles_per_16x16x32\": %.1f}\n", m + 1, t[m], flops / (t[m] * 1e-3) / 1e12,
               t[m] * 1e-3 * mhz * 1e6 / (iters * 32.0));
    // the same streams on U(-1,1) operands: which shape sustains more under the power limit?  (~0.5 s of burn-in each)
    {
        std::vector<uint16_t> h(1024 * 8);
        uint32_t lcg = 12345u;
        for (auto& v : h) {
            lcg = lcg * 1664525u + 1013904223u;
            const float x = 2.f * (float)(lcg >> 8) / 16777216.f - 1.f;
            uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); v = (uint16_t)(u >> 16);
        }
        hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const int it2 = 200000;
        const double fl2 = (double)cus * 4 * it2 * 16 * 2.0 * 32 * 32 * 16;
        for (int rep = 0; rep < 2; ++rep) {
            const double r0 = run_ms<0>(d, out, cus, it2), r2 = run_ms<2>(d, out, cus, it2), r5 = run_ms<5>(d, out, cus, it2);
            printf("{\"data\": \"U(-1,1)\", \"tflops_32x32x16\": %.0f, \"tflops_16x16x32\": %.0f, \"tflops_16x16x32_two_operand_sets\": %.0f}\n",
                   fl2 / (r0 * 1e-3) / 1e12, fl2 / (r2 * 1e-3) / 1e12, fl2 / (r5 * 1e-3) / 1e12);
        }
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("# error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
