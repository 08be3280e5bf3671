// Microbenchmark (tools only): what does handing the matrix pipe from one wave to the other wave of the same SIMD cost?
// 8 waves per CU (two per SIMD) as two "rows" of four, nothing but v_mfma_f32_32x32x16_bf16 (8 independent accumulators per
// wave, zero-filled data) and workgroup barriers, in the cadence of the 16-bit GETT kernel:
//     per phase and row:  s_barrier ; N1 MFMAs ; s_barrier ; N2 MFMAs        (row 1 runs one barrier behind row 0)
//   N1 = 8, N2 = 0 : the kernel's ping-pong skeleton (one row computes, the other waits at the barrier)
//   N1 = 6, N2 = 2 : the computing row releases the other one two MFMAs before the end of its segment
//   N1 = 4, N2 = 4 : both rows always have MFMAs to issue, a barrier every 4
//   N1 = 16, N2 = 0: segments twice as long
// plus the same MFMA count with no barriers at all on 8 waves (two free-running waves per SIMD) and on 4 waves (one per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_handover.hip -o tools/ubench/mfma_handover
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short  s16x8 __attribute__((ext_vector_type(8)));
typedef float  f32x16 __attribute__((ext_vector_type(16)));

// NACC accumulators per segment: 8 = all independent; 2 = the GETT kernel's order (two fragments of one quadrant, four k-steps each:
// every MFMA depends on the one two before it), the quadrant changing from phase to phase
#define MFMA2(PH, I) acc[2 * ((PH) & 3) + ((I) & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[((I) >> 1) & 3]), __builtin_bit_cast(bf16x8, b[((I) >> 1) & 3]), acc[2 * ((PH) & 3) + ((I) & 1)], 0, 0, 0);
#define MFMA(I) acc[(I) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(I) & 3]), __builtin_bit_cast(bf16x8, b[((I) >> 1) & 3]), acc[(I) & 7], 0, 0, 0);
#define FENCE() asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));

// MODE 0: barriers as described; MODE 1: no barriers.  PRIO: s_setprio(1) around the MFMA runs.
template <int THREADS, int N1, int N2, int MODE, int PRIO, int NACC = 8>
__global__ void __launch_bounds__(THREADS, 1) k(const s16x8* __restrict__ data, float* out, int phases) {
    s16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = data[(2 * i) * 64 + (threadIdx.x & 63)]; b[i] = data[(2 * i + 1) * 64 + (threadIdx.x & 63)]; }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int row = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    if (MODE == 0 && row == 1) __builtin_amdgcn_s_barrier();
    if constexpr (NACC == 2) {
        for (int ph4 = 0; ph4 < phases; ph4 += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE == 0) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < N1; ++i) { MFMA2(q, i) }
                asm volatile("" : "+v"(acc[2 * q]), "+v"(acc[2 * q + 1]));
                if (PRIO) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 0) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    for (int ph = 0; ph < phases; ++ph) {
        if (MODE == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < N1; ++i) { MFMA(i) }
        FENCE()
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (N2 > 0) {
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < N2; ++i) { MFMA(N1 + i) }
            FENCE()
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 0 && row == 0) __builtin_amdgcn_s_barrier();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][7];
    out[blockIdx.x * THREADS + threadIdx.x] = r;
}

template <int THREADS, int N1, int N2, int MODE, int PRIO, int NACC = 8>
void point(const char* name, const s16x8* d, float* out, int cus) {
    const int phases = 160000 / (N1 + N2);
    const double flops = (double)cus * (THREADS / 64) * phases * (N1 + N2) * 2.0 * 32 * 32 * 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k<THREADS, N1, N2, MODE, PRIO, NACC><<<cus, THREADS>>>(d, out, phases);
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) k<THREADS, N1, N2, MODE, PRIO, NACC><<<cus, THREADS>>>(d, out, phases);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("{\"variant\": \"%s\", \"waves\": %d, \"accumulators_per_segment\": %d, \"n1\": %d, \"n2\": %d, \"barriers\": %d, \"setprio\": %d, \"tflops\": %.1f}\n",
           name, THREADS / 64, NACC, N1, N2, MODE == 0, PRIO, 3.0 * flops / (ms * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    s16x8* d; float* out;
    hipMalloc(&d, 8 * 64 * sizeof(s16x8)); hipMemset(d, 0, 8 * 64 * sizeof(s16x8));
    hipMalloc(&out, (size_t)cus * 512 * sizeof(float));
    point<512, 8, 0, 1, 0>("free, two waves per SIMD", d, out, cus);
    point<512, 8, 0, 0, 0>("ping-pong 8", d, out, cus);
    point<512, 8, 0, 0, 1>("ping-pong 8, setprio", d, out, cus);
    point<512, 7, 1, 0, 0>("ping-pong 7 + 1 after the release", d, out, cus);
    point<512, 6, 2, 0, 0>("ping-pong 6 + 2 after the release", d, out, cus);
    point<512, 6, 2, 0, 1>("ping-pong 6 + 2 after the release, setprio", d, out, cus);
    point<512, 4, 4, 0, 0>("barrier every 4, both rows busy", d, out, cus);
    point<512, 16, 0, 0, 0>("ping-pong 16", d, out, cus);
    point<512, 14, 2, 0, 0>("ping-pong 14 + 2 after the release", d, out, cus);
    point<512, 12, 4, 0, 0>("ping-pong 12 + 4 after the release", d, out, cus);
    point<512, 32, 0, 0, 0>("ping-pong 32", d, out, cus);
    point<512, 8, 0, 0, 0, 2>("ping-pong 8 on two accumulators (kernel order)", d, out, cus);
    point<512, 8, 0, 0, 1, 2>("ping-pong 8 on two accumulators (kernel order), setprio", d, out, cus);
    point<512, 8, 0, 1, 0, 2>("free, two waves per SIMD, two accumulators per 8", d, out, cus);
    return 0;
}
