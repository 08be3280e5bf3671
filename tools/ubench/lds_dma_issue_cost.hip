// Microbenchmark (tools only, not part of the library; written at the end of round 1 — first measurements are the opening
// move of the next round): what does one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) cost the
// matrix pipe, depending on WHO issues it?  DESIGN.md section 6 found ~17 cycles per piece when the wave that issues the
// MFMAs also issues the pieces (four-wave bf16 kernel).  Modes, per CU one workgroup:
//   0: 4 waves (one per SIMD) issue nothing but independent v_mfma_f32_32x32x16_bf16            (reference rate)
//   1: the same 4 waves also issue one piece every PERIOD MFMAs                                 (self-issue)
//   2: 8 waves: waves 0-3 only MFMAs, waves 4-7 (one per SIMD, VALU-free) issue the same number of pieces (dedicated issuers)
// Every piece re-reads the same 64 KiB per CU (L2-resident), so the experiment measures issue cost, not bandwidth.
// Prints MFMA TFLOP/s per mode; build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_issue_cost.hip -o /tmp/dma_cost
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float  f32x16 __attribute__((ext_vector_type(16)));
typedef int    rsrc_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma_piece(rsrc_t r, uint32_t laneBytes, uint32_t ldsByte) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(ldsByte), "v"(laneBytes), "s"(r) : "memory");
}

// the same 1 KiB as four dword pieces (buffer_load_dword ... lds: 256 B per wave-instruction)
__device__ __forceinline__ void dma_piece_b32x4(rsrc_t r, uint32_t laneBytes, uint32_t ldsByte) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(ldsByte + 256u * q), "v"(laneBytes / 4u + 256u * q), "s"(r) : "memory");
}
// one dword piece (a quarter of the bytes), for modes that spread the four quarters over the MFMA stream
__device__ __forceinline__ void dma_piece_b32(rsrc_t r, uint32_t laneBytes, uint32_t ldsByte) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(ldsByte), "v"(laneBytes), "s"(r) : "memory");
}

template <int MODE, int PERIOD>
__global__ void __launch_bounds__(512, 1) k(const char* src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t base = (uint64_t)(uintptr_t)(src + (size_t)blockIdx.x * 65536);
    rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(base >> 32) & 0xffffu));
    r[2] = -1;
    r[3] = 0x00020000;
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    float res = 0.f;
    if (wave < 4) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.5f - 0.003f * (lane ^ i)); }
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        uint32_t piece = (uint32_t)wave * 16u;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 7], 0, 0, 0);
                if (MODE == 1 && (i % PERIOD) == PERIOD - 1) {
                    dma_piece(r, (uint32_t)lane * 16u + (piece & 63u) * 1024u, ldsBase + (piece & 63u) * 1024u);
                    ++piece;
                }
                if (MODE == 3 && (i % PERIOD) == PERIOD - 1) {
                    dma_piece_b32x4(r, (uint32_t)lane * 16u + (piece & 63u) * 1024u, ldsBase + (piece & 63u) * 1024u);
                    ++piece;
                }
                if (MODE == 4 && PERIOD >= 4 && (i % (PERIOD / 4)) == (PERIOD / 4) - 1) {   // same bytes per MFMA, spread out
                    dma_piece_b32(r, (uint32_t)lane * 4u + (piece & 255u) * 256u, ldsBase + (piece & 255u) * 256u);
                    ++piece;
                }
            }
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (MODE == 3 || MODE == 4) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        }
        for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][9];
    } else if (MODE == 2) {
        uint32_t piece = (uint32_t)(wave - 4) * 16u;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16 / PERIOD; ++i) {
                dma_piece(r, (uint32_t)lane * 16u + (piece & 63u) * 1024u, ldsBase + (piece & 63u) * 1024u);
                ++piece;
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_sleep(8);   // pace the issuer roughly like a 16-MFMA step of its SIMD's compute wave
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave < 4) out[blockIdx.x * 256 + threadIdx.x] = res + lds[lane];
}

template <int MODE, int PERIOD>
void run(const char* what, const char* src, float* out, int cus) {
    const int iters = 20000;
    const int threads = MODE == 2 ? 512 : 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) k<MODE, PERIOD><<<cus, threads>>>(src, out, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) k<MODE, PERIOD><<<cus, threads>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 3.0 * cus * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("{\"mode\": \"%s\", \"mfmas_per_piece\": %d, \"ms_per_launch\": %.3f, \"mfma_tflops\": %.0f}\n", what, PERIOD, ms / 3, flops / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    char* src; float* out;
    hipMalloc(&src, (size_t)cus * 65536); hipMemset(src, 0x3c, (size_t)cus * 65536);
    hipMalloc(&out, (size_t)cus * 256 * 4);
    run<0, 4>("mfma only", src, out, cus);
    run<1, 4>("self-issued pieces", src, out, cus);
    run<1, 2>("self-issued pieces", src, out, cus);
    run<1, 1>("self-issued pieces", src, out, cus);
    run<3, 4>("self-issued, 4 dword pieces back to back per KiB", src, out, cus);
    run<3, 2>("self-issued, 4 dword pieces back to back per KiB", src, out, cus);
    run<4, 4>("self-issued, one dword piece per MFMA (1 KiB per 4 MFMAs)", src, out, cus);
    run<4, 8>("self-issued, one dword piece per 2 MFMAs (1 KiB per 8 MFMAs)", src, out, cus);
    run<1, 8>("self-issued pieces", src, out, cus);
    run<2, 4>("dedicated issuer waves", src, out, cus);
    run<2, 2>("dedicated issuer waves", src, out, cus);
    run<2, 1>("dedicated issuer waves", src, out, cus);
    return 0;
}
