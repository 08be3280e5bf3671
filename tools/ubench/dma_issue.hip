// Microbenchmark (tools only): what does ONE wave per SIMD pay for an LDS-DMA instruction (buffer_load_dwordx4 ... lds) issued
// between its MFMAs?  Four waves per workgroup, one workgroup per CU, zero operands (no power limit in the way), a loop of
//     { N x 32 MFMA-cycles ; one memory instruction }
// for N = 1, 2, 4, 8 and both bf16 MFMA shapes (32x32x16: 8 passes, 16x16x32: 4 passes — two of them per "32 cycles").
// Modes:  0  no memory instruction (the MFMA-only time the others are compared with)
//         1  LDS-DMA, M0 written once in front of the loop
//         2  s_mov_b32 m0 + s_nop 0 + LDS-DMA (what the GETT kernels do), four LDS slots in turn
//         3  buffer_load_dwordx4 into registers (no LDS), four register sets in turn, never waited for
//         4  LDS-DMA of ONE dword per lane (256 B per instruction), M0 constant
//         5  mode 2 behind s_waitcnt vmcnt(6) (at most seven pieces in flight per wave)
//         6  global_load_lds_dwordx4 (flat-global addressing instead of a buffer descriptor), M0 per instruction
// The source is 64 KiB per workgroup (L2-resident after the first pass).  Reported per configuration: ns per loop iteration, and the
// extra SIMD cycles per memory instruction = (t / t_mode0 - 1) * N * 32.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_issue.hip -o tools/ubench/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSlotBytes = 4096;       // four waves x 1 KiB

template <int SHAPE, int N, int MODE>
__global__ void __launch_bounds__(256, 1) probe(const char* __restrict__ src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[8 * kSlotBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * 65536;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 65536, 0x00020000);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (uint32_t)wave * 1024u;
    uint32_t slot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) slot[i] = __builtin_amdgcn_readfirstlane(ldsBase + i * kSlotBytes);
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) voff[i] = (uint32_t)(lane * 16 + wave * 1024 + i * 16384);
    const uint32_t voff4 = (uint32_t)(lane * 4 + wave * 256);
    uint64_t gaddr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gaddr[i] = (uint64_t)(uintptr_t)(base + voff[i]);

    s16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    a[0] = (short)(src[lane] & 0);           // opaque zeros
    b[0] = (short)(src[lane + 64] & 0);
    f32x16 acc32[4];
    f32x4 acc16[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 regs[4] = {};

    if (MODE == 1 || MODE == 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(slot[0]) : "memory");

    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {        // four { N MFMA slots ; memory instruction } groups per trip
#pragma unroll
            for (int m = 0; m < N; ++m) {
                if constexpr (SHAPE == 0) {
                    acc32[(u * N + m) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                                                   acc32[(u * N + m) & 3], 0, 0, 0);
                } else {
                    acc16[(2 * (u * N + m)) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc16[(2 * (u * N + m)) & 7], 0, 0, 0);
                    acc16[(2 * (u * N + m) + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc16[(2 * (u * N + m) + 1) & 7], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE == 1) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff[u]), "s"(rsrc) : "memory");
            if constexpr (MODE == 2)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(slot[u]), "v"(voff[u]), "s"(rsrc) : "memory");
            if constexpr (MODE == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(regs[u]) : "v"(voff[u]), "s"(rsrc) : "memory");
            if constexpr (MODE == 4) asm volatile("buffer_load_dword %0, %1, 0 offen lds" ::"v"(voff4), "s"(rsrc) : "memory");
            if constexpr (MODE == 5)
                asm volatile("s_waitcnt vmcnt(6)\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(slot[u]), "v"(voff[u]), "s"(rsrc)
                             : "memory");
            if constexpr (MODE == 6)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(slot[u]), "v"(gaddr[u]) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) r += acc32[i][0] + acc32[i][9];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc16[i][1];
    if constexpr (MODE == 3) r += (float)(regs[0][0] ^ regs[1][1] ^ regs[2][2] ^ regs[3][3]);
    __syncthreads();
    r += (float)lds[threadIdx.x * 16];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

static char* gSrc;
static float* gOut;
static int gCus, gMask = 127;

template <int SHAPE, int N, int MODE>
static double run_ns(int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) probe<SHAPE, N, MODE><<<gCus, 256>>>(gSrc, gOut, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) probe<SHAPE, N, MODE><<<gCus, 256>>>(gSrc, gOut, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    fprintf(stderr, "# shape %d N %d mode %d: %.3f ms\n", SHAPE, N, MODE, ms);
    return (double)ms * 1e6 / 3.0 / ((double)iters * 4.0);     // ns per { N MFMA slots ; instruction } group
}

template <int SHAPE, int N>
static void row(int iters) {
    const double t0 = run_ns<SHAPE, N, 0>(iters);
    double t[6] = {t0, t0, t0, t0, t0, t0};
    if (gMask & 2) t[0] = run_ns<SHAPE, N, 1>(iters);
    if (gMask & 4) t[1] = run_ns<SHAPE, N, 2>(iters);
    if (gMask & 8) t[2] = run_ns<SHAPE, N, 3>(iters);
    if (gMask & 16) t[3] = run_ns<SHAPE, N, 4>(iters);
    if (gMask & 32) t[4] = run_ns<SHAPE, N, 5>(iters);
    if (gMask & 64) t[5] = run_ns<SHAPE, N, 6>(iters);
    printf("{\"mfma\": \"%s\", \"mfma_slots_per_instruction\": %d, \"ns_mfma_only\": %.2f, \"mhz_implied\": %.0f", SHAPE == 0 ? "32x32x16" : "16x16x32", N, t0,
           N * 32.0 / t0 * 1e3);
    const char* names[6] = {"dma_m0_const", "dma_m0_per_inst", "load_to_regs", "dma_dword", "dma_vmcnt6", "global_load_lds"};
    for (int i = 0; i < 6; ++i) printf(", \"extra_cycles_%s\": %.1f", names[i], (t[i] / t0 - 1.0) * N * 32.0);
    printf("}\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) gMask = atoi(argv[1]);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    gCus = prop.multiProcessorCount;
    hipMalloc(&gSrc, (size_t)gCus * 65536);
    hipMemset(gSrc, 0, (size_t)gCus * 65536);
    hipMalloc(&gOut, (size_t)gCus * 256 * 4);
    const int iters = 20000;
    row<0, 1>(iters); row<0, 2>(iters); row<0, 4>(iters); row<0, 8>(iters);
    row<1, 1>(iters); row<1, 2>(iters); row<1, 4>(iters); row<1, 8>(iters);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("# error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
