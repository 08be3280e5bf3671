// Microbenchmark (tools only, not part of the library): sustained bf16 MFMA rate of gfx950 under its power limit as a
// function of (a) the MFMA shape — v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16, same flops per cycle on paper —
// (b) the operand data — zeros, a constant, U(-1,1) — and (c) how many distinct operand registers the stream cycles
// through.  Every CU runs one 256-thread workgroup (one wave per SIMD) issuing nothing but independent MFMAs for long
// enough (~100 ms per point) to sit in the steady state.  Prints TFLOP/s per point.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shape_power.hip -o /tmp/mfma_shape_power && /tmp/mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short  s16x8 __attribute__((ext_vector_type(8)));
typedef float  f32x16 __attribute__((ext_vector_type(16)));
typedef float  f32x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: 32x32x16 (16 accumulator registers per MFMA), 1: 16x16x32 (4 registers).  NOPS distinct A and B registers.
template <int SHAPE, int NOPS>
__global__ void __launch_bounds__(256, 1) k(const s16x8* __restrict__ data, float* out, int iters) {
    s16x8 a[NOPS], b[NOPS];
#pragma unroll
    for (int i = 0; i < NOPS; ++i) {
        a[i] = data[(i * 2 + 0) * 256 + threadIdx.x];
        b[i] = data[(i * 2 + 1) * 256 + threadIdx.x];
    }
    float r = 0.f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i % NOPS]), __builtin_bit_cast(bf16x8, b[(i / 2) % NOPS]), acc[i & 7], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][7];
    } else {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i)     // 32 x (16x16x32) = the flops of 16 x (32x32x16)
                acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i % NOPS]), __builtin_bit_cast(bf16x8, b[(i / 2) % NOPS]), acc[i & 15], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

static uint16_t bf16_of(float f) { uint32_t u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

template <int SHAPE, int NOPS>
void point(const char* what, const s16x8* d, float* out, int cus) {
    const int iters = 40000;
    const double flops = (double)cus * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k<SHAPE, NOPS><<<cus, 256>>>(d, out, iters);      // burn-in (~2 x 25-50 ms)
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) k<SHAPE, NOPS><<<cus, 256>>>(d, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("{\"shape\": \"%s\", \"operand_regs\": %d, \"data\": \"%s\", \"ms_per_launch\": %.3f, \"tflops\": %.0f}\n",
           SHAPE == 0 ? "32x32x16" : "16x16x32", NOPS, what, ms / 3, 3 * flops / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t n = 16 * 256 * 8;     // 8 operand pairs x 256 lanes x 8 bf16
    std::vector<uint16_t> h(n);
    s16x8* d; float* out;
    hipMalloc(&d, n * 2); hipMalloc(&out, (size_t)cus * 256 * 4);
    const char* names[3] = {"zeros", "constant 0.75", "U(-1,1)"};
    for (int kind = 0; kind < 3; ++kind) {
        srand(7);
        for (size_t i = 0; i < n; ++i)
            h[i] = kind == 0 ? 0 : kind == 1 ? bf16_of(0.75f) : bf16_of(2.f * rand() / (float)RAND_MAX - 1.f);
        hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
        point<0, 1>(names[kind], d, out, cus);
        point<0, 8>(names[kind], d, out, cus);
        point<1, 1>(names[kind], d, out, cus);
        point<1, 8>(names[kind], d, out, cus);
    }
    return 0;
}
