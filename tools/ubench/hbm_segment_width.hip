// Microbenchmark (tools only): what does the WIDTH of the contiguous row segment do to an LDS-DMA operand stream that comes out of
// HBM?  The headline einsum's A operand is [96 rows][K = 262144] fp32, K-contiguous, rows 1 MiB apart; every workgroup owns a
// K-slice of 1024 floats and stages it K-tile by K-tile: with the kernel's K-tile of 32 a row contributes 128 contiguous bytes per
// K-tile (a 1-KiB LDS-DMA piece = 8 rows x 128 B), with a K-tile of 64 / 128 it would be 256 B / 512 B (4 rows / 2 rows per piece).
// Same bytes, same 1-KiB pieces, same bytes in flight — only the segment width differs.  The cold headline kernel is bound by this
// stream (profiles/r05i_headline_cold_counters.json: the stream alone 41.3 us for 201 MB = 4.9 TB/s), so this is what a deeper K-tile
// could change in the HBM-cold case.
// 256 workgroups (one per CU) of 4 waves, each wave at most Q pieces in flight (4 Q KiB per CU; the kernel's ring of 3 holds 48 KiB),
// LDS ring of 64 KiB, nothing reads it.  SEG = 0: the B operand's pattern for comparison — a K-tile is ONE contiguous 12-KiB block.
// Operand copies rotate (8 x 100.7 MB = 805 MB > the 256-MiB Infinity Cache) or do not (warm: one copy, re-read).
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/hbm_segment_width.hip -o tools/ubench/hbm_segment_width
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRows = 96, kSlice = 1024;            // floats of K per workgroup
constexpr size_t kK = 262144;                       // floats per row
constexpr int kPieces = kRows * kSlice / 256;       // 1-KiB pieces per workgroup: 384

template <int SEG, int Q, int NT>                   // SEG floats per row segment (32 / 64 / 128; 0 = contiguous tiles), NT: nontemporal
__global__ void __launch_bounds__(256, 1) stream(const float* __restrict__ A) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, -1, 0x00020000);
    const uint32_t k0 = blockIdx.x * (uint32_t)kSlice;
    for (int i = 0; i < kPieces / 4; ++i) {
        const int p = 4 * i + wave;                 // this wave's piece; the four waves walk neighbouring pieces
        uint32_t off;
        if constexpr (SEG == 0) {
            // B-like: the workgroup's 384 KiB are one contiguous range (a K-tile = 12 KiB in a row)
            off = (uint32_t)(((size_t)blockIdx.x * kRows * kSlice + (size_t)p * 256 + lane * 4) * 4);
        } else {
            constexpr int lanesPerRow = SEG / 4, rowsPerPiece = 64 / lanesPerRow, piecesPerTile = kRows / rowsPerPiece;
            const int tile = p / piecesPerTile, q = p % piecesPerTile;
            const int row = q * rowsPerPiece + lane / lanesPerRow;
            const uint32_t col = k0 + (uint32_t)tile * SEG + (uint32_t)(lane % lanesPerRow) * 4u;
            off = (uint32_t)(((size_t)row * kK + col) * 4);
        }
        const uint32_t slot = __builtin_amdgcn_readfirstlane(ldsBase + (uint32_t)(((4 * i + wave) & 63) * 1024));
        if constexpr (NT)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(slot), "v"(off), "s"(rsrc) : "memory");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(slot), "v"(off), "s"(rsrc) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int SEG, int Q, int NT>
static void run(const char* what, std::vector<float*>& copies, bool cold) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 400;
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((stream<SEG, Q, NT>), dim3(256), dim3(256), 0, 0, copies[cold ? i % copies.size() : 0]);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream<SEG, Q, NT>), dim3(256), dim3(256), 0, 0, copies[cold ? i % copies.size() : 0]);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)kRows * kK * 4.0;
    printf("{\"pattern\": \"%s\", \"segment_bytes\": %d, \"pieces_in_flight_per_wave\": %d, \"nontemporal\": %d, \"operands\": \"%s\", \"us_per_launch\": %.2f, \"TBps\": %.3f}\n",
           what, SEG * 4, Q, NT, cold ? "cold (8 rotating copies, 805 MB)" : "warm (one copy)", us, bytes / (us * 1e-6) / 1e12);
    fflush(stdout);
}

int main() {
    std::vector<float*> copies(8);
    const size_t n = (size_t)kRows * kK;
    for (auto& c : copies) {
        if (hipMalloc(&c, n * 4) != hipSuccess) { printf("{\"error\": \"hipMalloc\"}\n"); return 1; }
        hipMemset(c, 0x3c, n * 4);
    }
    hipDeviceSynchronize();
    // clock ramp
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL((stream<32, 12, 0>), dim3(256), dim3(256), 0, 0, copies[i % 8]);
    hipDeviceSynchronize();
    for (int cold = 1; cold >= 0; --cold) {
        run<32, 12, 0>("A-like rows", copies, cold);     // the kernel's K-tile of 32: 48 KiB in flight per CU
        run<64, 12, 0>("A-like rows", copies, cold);
        run<128, 12, 0>("A-like rows", copies, cold);
        run<0, 12, 0>("B-like contiguous tiles", copies, cold);
        run<32, 12, 1>("A-like rows", copies, cold);
        run<64, 12, 1>("A-like rows", copies, cold);
        run<128, 12, 1>("A-like rows", copies, cold);
        run<0, 12, 1>("B-like contiguous tiles", copies, cold);
        run<32, 24, 0>("A-like rows", copies, cold);     // twice the bytes in flight
        run<128, 24, 0>("A-like rows", copies, cold);
        run<0, 24, 0>("B-like contiguous tiles", copies, cold);
    }
    return 0;
}
