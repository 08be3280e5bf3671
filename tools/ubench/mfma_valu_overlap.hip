// Microbenchmark (tools only, not part of the library): do fp32 MFMA (v_mfma_f32_16x16x4_f32) and fp32
// VALU FMA (v_pk_fma_f32) issue concurrently on one SIMD of gfx950, i.e. is the f32 matrix rate additive
// with the f32 vector rate?  8 waves per CU: waves 0-3 run MFMA chains, waves 4-7 run packed-FMA chains.
// mode 0: MFMA waves only, 1: VALU waves only, 2: both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(float* out, unsigned long long* clk, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    float r = 0.f;
    if (wave < 4) {
        if (MODE != 1) {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else {
        if (MODE != 0) {
            f32x2 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 0.f};
            f32x2 a = {seed + threadIdx.x * 1e-3f, seed}, b = {seed * 0.5f, seed + threadIdx.x * 2e-3f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
            for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0; clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}

template <int MODE>
void run(const char* name, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(out, clk, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 512>>>(out, clk, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512]; hipMemcpy(h, clk, 256 * 16, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0; for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    const double mfmaFlop = (MODE != 1) ? 256.0 * 4 * iters * 8 * 2048.0 : 0;     // 16*16*4*2 per MFMA
    const double valuFlop = (MODE != 0) ? 256.0 * 4 * iters * 16 * 256.0 : 0;     // 64 lanes * 2 * 2 per pk_fma
    printf("%s: %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF  clock %.3f GHz  cycles/iter %.1f\n", name, ms,
           mfmaFlop / ms / 1e9, valuFlop / ms / 1e9, (mfmaFlop + valuFlop) / ms / 1e9, cyc / (wall * 10.0), cyc / 256 / iters);
    hipFree(out); hipFree(clk);
}
int main() {
    const int iters = 20000;
    run<0>("mfma only", iters);
    run<1>("valu only", iters);
    run<2>("both     ", iters);
    return 0;
}
