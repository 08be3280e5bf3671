// Microbenchmark (tools only): does the sustained v_mfma_f32_32x32x16_bf16 rate of one wave per SIMD depend on how many
// accumulator tiles the stream cycles through (8 = 128 registers, 16 = 256 registers: the 4-wave GETT kernel's 128 x 128
// wave tile) and on whether consecutive MFMAs share an A operand register?  Prints TFLOP/s per point (zero-filled data).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_acc_count.hip -o tools/ubench/mfma_acc_count
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short  s16x8 __attribute__((ext_vector_type(8)));
typedef float  f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int SHARE>
__global__ void __launch_bounds__(256, 1) k(const s16x8* __restrict__ data, float* out, int iters) {
    s16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = data[(2 * i) * 256 + threadIdx.x]; b[i] = data[(2 * i + 1) * 256 + threadIdx.x]; }
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ai = SHARE ? (i >> 2) : (i & 3), bi = SHARE ? (i & 3) : ((i >> 2) & 3);
            acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ai]), __builtin_bit_cast(bf16x8, b[bi]), acc[i % NACC], 0, 0, 0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][7];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NACC, int SHARE>
void point(const s16x8* d, float* out, int cus) {
    const int iters = 20000;
    const double flops = (double)cus * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k<NACC, SHARE><<<cus, 256>>>(d, out, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) k<NACC, SHARE><<<cus, 256>>>(d, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("{\"accumulators\": %d, \"a_operand_shared_by_4_consecutive\": %d, \"tflops\": %.0f}\n", NACC, SHARE, 3 * flops / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    s16x8* d; float* out;
    hipMalloc(&d, 8 * 256 * 16); hipMemset(d, 0, 8 * 256 * 16);
    hipMalloc(&out, (size_t)cus * 256 * 4);
    point<8, 0>(d, out, cus); point<8, 1>(d, out, cus); point<16, 0>(d, out, cus); point<16, 1>(d, out, cus); point<4, 1>(d, out, cus);
    return 0;
}
