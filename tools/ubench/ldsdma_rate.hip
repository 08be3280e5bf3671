// Microbenchmark (tools only): how many bytes per clock does ONE CU pull into LDS through buffer_load_dwordx4 ... lds when it does
// nothing else?  Every 16-bit LDS-DMA GETT kernel of this repository ends up at 28-36 B/clk/CU (256 x 256 tile: 64 KiB per 2277
// cycles; 128 x 128 tile: 32 KiB per ~930) where the texture path is quoted at 64 B/clk — is that the path's practical rate from
// L2, or do the kernels leave it on the table?  One workgroup per CU, W waves, every wave issues 1-KiB pieces back to back into a
// ring of LDS slots with at most Q pieces in flight (s_waitcnt vmcnt(Q - 1) behind each issue).  Source footprint per workgroup F:
//   64 KiB   (re-read all the time: L1 / L2 resident)
//   2 MiB    (32 CUs x 2 MiB = 64 MiB per XCD: beyond L2, inside the 256-MiB Infinity Cache when SHARE = 1 below)
//   SHARE    the 32 workgroups of an XCD read the SAME 2 MiB (operand panels shared by the tiles of an XCD: L2 hits)
// Reported per configuration: bytes per shader clock per CU (s_memtime inside the kernel) and GB/s per CU / TB/s chip (events).
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/ldsdma_rate.hip -o tools/ubench/ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

template <int W, int Q>
__global__ void __launch_bounds__(64 * W, 1) stream(const char* __restrict__ src, unsigned long long* stamps, uint32_t footprint, uint32_t share, int pieces) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // SHARE: workgroup b runs on XCD b % 8 (observed): the workgroups of an XCD share one footprint
    const uint32_t region = share ? (blockIdx.x & 7u) : blockIdx.x;
    const char* base = src + (size_t)region * footprint;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)footprint, 0x00020000);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t mask = footprint - 1u;              // power of two
    uint32_t off = (uint32_t)(wave * 1024 + lane * 16) + (share ? (blockIdx.x >> 3) * 4096u * W : 0u);   // sharers start at different places
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < pieces; ++i) {
        const uint32_t slot = __builtin_amdgcn_readfirstlane(ldsBase + (uint32_t)((wave * 8 + (i & 7)) * 1024) % (64u * 1024u));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(slot), "v"(off & mask), "s"(rsrc) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q - 1) : "memory");
        off += (uint32_t)(W * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

// The same stream next to LDS READERS: waves 0-3 move data as above (16 pieces in flight each), waves 4-7 read the ring with
// ds_read_b128 — RPP reads per piece-time, i.e. RPP KiB read per KiB staged (a 128 x 128 x 64 GETT tile reads 2 bytes of fragments
// per byte it stages, the 256 x 256 tile also 2).  Does the DMA rate hold?  (If not, LDS-DMA writes and fragment reads share the LDS
// pipeline and a GETT kernel is bound by their SUM, not by either.)
template <int RPP>
__global__ void __launch_bounds__(512, 1) stream_with_readers(const char* __restrict__ src, unsigned long long* stamps, float* sink, uint32_t footprint, int pieces) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3;
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    if (wave8 >= 4) {          // readers: RPP x 1 KiB per piece the movers stage, for as long as the movers run (same trip count)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const char* base = lds + lane * 16;
        for (int i = 0; i < pieces; ++i) {
#pragma unroll
            for (int r = 0; r < RPP; ++r) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + (((i * RPP + r) * 4 + wave) & 63) * 1024);
                acc += v;
            }
        }
        if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[threadIdx.x] = acc[0];
        return;
    }
    const char* base = src + (size_t)blockIdx.x * footprint;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)footprint, 0x00020000);
    const uint32_t mask = footprint - 1u;
    uint32_t off = (uint32_t)(wave * 1024 + lane * 16);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < pieces; ++i) {
        const uint32_t slot = __builtin_amdgcn_readfirstlane(ldsBase + (uint32_t)((wave * 16 + (i & 15)) * 1024));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(slot), "v"(off & mask), "s"(rsrc) : "memory");
        asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        off += 4096u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

template <int RPP>
static void run_readers(const char* src, unsigned long long* stamps, float* sink, int cus) {
    const int pieces = 8192;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((stream_with_readers<RPP>), dim3(cus), dim3(512), 0, nullptr, src, stamps, sink, 64u << 10, pieces);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * (size_t)cus);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0.0;
    for (int b = 0; b < cus; ++b) cyc += (double)(h[2 * b + 1] - h[2 * b]);
    cyc /= cus;
    const double bytesWg = (double)pieces * 4 * 1024.0;
    printf("{\"source\":\"64 KiB per workgroup, 4 movers + 4 readers\",\"KiB_read_per_KiB_staged\":%d,\"dma_bytes_per_clk_per_cu\":%.1f,\"lds_read_bytes_per_clk_per_cu\":%.1f}\n", RPP,
           bytesWg / cyc, bytesWg * RPP / cyc);
}

template <int W, int Q>
static void run(const char* src, unsigned long long* stamps, int cus, uint32_t footprint, uint32_t share, const char* what) {
    const int pieces = 8192;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream<W, Q>), dim3(cus), dim3(64 * W), 0, nullptr, src, stamps, footprint, share, pieces);
    hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL((stream<W, Q>), dim3(cus), dim3(64 * W), 0, nullptr, src, stamps, footprint, share, pieces);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * (size_t)cus);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0.0;
    for (int b = 0; b < cus; ++b) cyc += (double)(h[2 * b + 1] - h[2 * b]);
    cyc /= cus;
    const double bytesWg = (double)pieces * W * 1024.0;
    printf("{\"source\":\"%s\",\"waves\":%d,\"in_flight_per_wave\":%d,\"bytes_per_clk_per_cu\":%.1f,\"GBps_per_cu\":%.1f,\"TBps_chip\":%.2f,\"ms\":%.3f}\n", what, W, Q,
           bytesWg / cyc, bytesWg / (ms * 1e-3) / 1e9, bytesWg * cus / (ms * 1e-3) / 1e12, ms);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t bytes = (size_t)cus * (2u << 20);
    char* src = nullptr;
    unsigned long long* stamps = nullptr;
    hipMalloc((void**)&src, bytes);
    hipMemset(src, 1, bytes);
    hipMalloc((void**)&stamps, 2 * (size_t)cus * 8);
    run<4, 8>(src, stamps, cus, 64u << 10, 0, "64 KiB per workgroup (cache resident)");
    run<4, 16>(src, stamps, cus, 64u << 10, 0, "64 KiB per workgroup (cache resident)");
    run<8, 8>(src, stamps, cus, 64u << 10, 0, "64 KiB per workgroup (cache resident)");
    run<8, 16>(src, stamps, cus, 64u << 10, 0, "64 KiB per workgroup (cache resident)");
    run<4, 16>(src, stamps, cus, 2u << 20, 1, "2 MiB shared by the 32 workgroups of an XCD (L2 hits)");
    run<8, 16>(src, stamps, cus, 2u << 20, 1, "2 MiB shared by the 32 workgroups of an XCD (L2 hits)");
    run<4, 16>(src, stamps, cus, 2u << 20, 0, "2 MiB per workgroup (512 MiB in all: HBM / Infinity Cache)");
    run<8, 16>(src, stamps, cus, 2u << 20, 0, "2 MiB per workgroup (512 MiB in all: HBM / Infinity Cache)");
    run<8, 32>(src, stamps, cus, 2u << 20, 0, "2 MiB per workgroup (512 MiB in all: HBM / Infinity Cache)");
    float* sink = nullptr;
    hipMalloc((void**)&sink, 4096);
    run_readers<1>(src, stamps, sink, cus);
    run_readers<2>(src, stamps, sink, cus);
    run_readers<4>(src, stamps, sink, cus);
    run_readers<8>(src, stamps, sink, cus);
    hipFree(sink);
    hipFree(src); hipFree(stamps);
    return 0;
}
