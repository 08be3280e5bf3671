// Microbenchmark (tools only): the LDS-DMA stream of a 128 x 128 x 64 16-bit GETT tile as the 32 workgroups of an XCD issue it —
// per K-step every workgroup stages 16 KiB of "its" A panel and 16 KiB of "its" B panel, the panels are SHARED (A panel ia by the 4
// workgroups with the same ia, B panel ib by 8), the sharers run in lockstep, and the 12 panels of an XCD (192 KiB per K-step,
// 48 MiB over the launch) stream through a 4-MiB L2 from the Infinity Cache.  The GETT kernels sit at ~74 GB/s per CU in this
// regime although a CU pulls 130 GB/s out of L2 (ldsdma_rate.hip) and only 19 % of the L2 requests miss.  Hypothesis: lockstep sharers
// all request a line while its miss is in flight and ALL wait out the miss; a DEDUPLICATED software prefetch (each workgroup touches
// its 1/4 of the A chunk and 1/8 of the B chunk of K-step t + D with one 4-byte load per line: 48 lanes of one instruction) turns the
// staging requests into true L2 hits.  Modes: 0 no prefetch, 1 deduplicated prefetch D steps ahead, 2 every workgroup prefetches all
// of its 256 lines.  Reported: microseconds per K-step (32 KiB per CU) and GB/s per CU.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/ldsdma_shared_panels.hip -o tools/ubench/ldsdma_shared_panels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

constexpr int kRing = 4;                    // K-steps of 32 KiB in LDS (128 KiB), kRing - 1 in flight

template <int MODE>
__global__ void __launch_bounds__(256, 1) panels(const char* __restrict__ src, unsigned long long* stamps, float* sink, int steps, int dist, int skew, int nA, int nB) {
    __shared__ __attribute__((aligned(16))) char lds[kRing * 32 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t j = blockIdx.x >> 3;                 // workgroup b runs on XCD b % 8 (observed): j = index inside the XCD
    const uint32_t ia = j % (uint32_t)nA, ib = (nB == 32) ? j : (j / (uint32_t)nA) % (uint32_t)nB;   // default 8 x 4: A panel 0..7, B panel 8..11
    const uint32_t chunk = 16u * 1024u;
    const size_t total = (size_t)(nA + nB) * steps * chunk;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)total, 0x00020000);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t offA = ia * (uint32_t)steps * chunk, offB = ((uint32_t)nA + ib) * (uint32_t)steps * chunk;
    // staging: piece i of wave w = KiB 4 w + i of the A chunk (i < 4) / of the B chunk (i >= 4)
    const uint32_t laneOff = (uint32_t)(wave * 4096 + lane * 16);
    // prefetch lanes: one line (128 B) per lane
    uint32_t pfOff = 0;
    bool pfOn = false;
    if (MODE == 1 && wave == 0) {
        if (lane < 32) { pfOff = offA + (32u * ib + (uint32_t)lane) * 128u; pfOn = true; }
        else if (lane < 48) { pfOff = offB + (16u * ia + (uint32_t)(lane - 32)) * 128u; pfOn = true; }
    }
    if (MODE == 2) {                                     // wave w: lines 64 w .. of the 256 (A: 0-127, B: 128-255)
        const uint32_t line = (uint32_t)(wave * 64 + lane);
        pfOff = (line < 128u) ? offA + line * 128u : offB + (line - 128u) * 128u;
        pfOn = true;
    }
    const int t00 = (skew != 0) ? (int)((ia + 8u * ib) % (uint32_t)skew) : 0;   // sharers start `skew` classes of K-steps apart (wrap)
    auto issue = [&](int t) {
        const int tt = (t + t00) % steps;
        const uint32_t buf = (uint32_t)(t % kRing) * 32768u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t slot = __builtin_amdgcn_readfirstlane(ldsBase + buf + (uint32_t)(i >= 4 ? 16384 : 0) + (uint32_t)(wave * 4096 + (i & 3) * 1024));
            // MODE 6 / 7: the sharers of a panel walk the 16 KiB of a chunk in ROTATED order (workgroup j starts 1 KiB x (j mod 16) into it):
            // at any instant the CUs of an XCD ask for different lines (L2 channels) of the shared chunk
            const uint32_t rot = (MODE == 6 || MODE == 7) ? (j * (MODE == 7 ? 5u : 1u)) & 15u : 0u;
            const uint32_t kib = ((uint32_t)(wave * 4 + (i & 3)) + rot) & 15u;
            const uint32_t go = (i >= 4 ? offB : offA) + (uint32_t)tt * chunk + kib * 1024u + (uint32_t)lane * 16u;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(slot), "v"(go), "s"(rsrc) : "memory");
            // MODE 4 / 5: self-clocked issue — behind every piece wait for the piece kRing - 1 K-steps older (the outstanding count
            // allowed is the same 8 (kRing - 2) at every piece), instead of a burst of eight behind the barrier
            if ((MODE == 4 || MODE == 5) && t >= kRing - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (kRing - 2)) : "memory");
        }
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < kRing - 1; ++t) issue(t);
    for (int t = 0; t < steps; ++t) {
        // K-step t has landed (the newer kRing - 2 steps stay in flight, plus this wave's prefetches issued since)
        if (MODE == 2 || (MODE == 1 && wave == 0)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9 * (kRing - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (kRing - 2)) : "memory");
        if (MODE != 3 && MODE != 5) __builtin_amdgcn_s_barrier();      // MODE 3: free-running waves (no workgroup barrier per K-step)
        if (MODE == 2 || (MODE == 1 && wave == 0)) {      // one 4-byte load per line, result never used (no wait is generated for it)
            const int tp = (t + t00 + dist) % steps;
            const uint32_t go = pfOff + (uint32_t)tp * chunk;
            if (pfOn) {
                uint32_t d;
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(d) : "v"(go), "s"(rsrc) : "memory");
            }
        }
        if (t + kRing - 1 < steps) issue(t + kRing - 1); else issue(steps - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

template <int MODE>
static void run(const char* src, unsigned long long* stamps, float* sink, int cus, int steps, int dist, int skew, int nA = 8, int nB = 4) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((panels<MODE>), dim3(cus), dim3(256), 0, nullptr, src, stamps, sink, steps, dist, skew, nA, nB);
    hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL((panels<MODE>), dim3(cus), dim3(256), 0, nullptr, src, stamps, sink, steps, dist, skew, nA, nB);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * (size_t)cus);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0.0;
    for (int b = 0; b < cus; ++b) cyc += (double)(h[2 * b + 1] - h[2 * b]);
    cyc /= cus;
    printf("{\"mode\":%d,\"panels_per_xcd\":%d,\"unique_KiB_per_k_step_per_xcd\":%d,\"steps\":%d,\"prefetch_distance\":%d,\"skew\":%d,\"us_per_k_step\":%.3f,\"cycles_per_k_step\":%.0f,\"GBps_per_cu\":%.1f,\"ms\":%.3f}\n", MODE, nA + nB, 16 * (nA + nB), steps, dist, skew,
           ms * 1e3 / steps, cyc / steps, 32768.0 * steps / (ms * 1e-3) / 1e9, ms);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int maxSteps = 256;
    const size_t bytes = (size_t)64 * maxSteps * 16384;
    char* src = nullptr;
    unsigned long long* stamps = nullptr;
    float* sink = nullptr;
    hipMalloc((void**)&src, bytes);
    hipMemset(src, 1, bytes);
    hipMalloc((void**)&stamps, 2 * (size_t)cus * 8);
    hipMalloc((void**)&sink, 4096);
    for (int steps : {32, 256}) {
        run<0>(src, stamps, sink, cus, steps, 0, 0);
        for (int d : {4, 8, 16}) run<1>(src, stamps, sink, cus, steps, d, 0);
        run<2>(src, stamps, sink, cus, steps, 8, 0);
        run<0>(src, stamps, sink, cus, steps, 0, 4);      // sharers 0..3 K-steps apart, no prefetch
        run<0>(src, stamps, sink, cus, steps, 0, 32);     // all 32 workgroups of an XCD on different K-steps
        // the sharing degree: 2 panels per XCD (everybody shares both) ... 64 (nobody shares): time per K-step against the unique
        // bytes an XCD pulls through its L2 per K-step
        run<0>(src, stamps, sink, cus, steps, 0, 0, 1, 1);
        run<3>(src, stamps, sink, cus, steps, 0, 0, 1, 1);   // the same without the barrier per K-step
        run<3>(src, stamps, sink, cus, steps, 0, 0, 8, 4);
        run<4>(src, stamps, sink, cus, steps, 0, 0, 1, 1);   // self-clocked issue, barrier per K-step
        run<4>(src, stamps, sink, cus, steps, 0, 0, 8, 4);
        run<5>(src, stamps, sink, cus, steps, 0, 0, 1, 1);   // self-clocked issue, no barrier
        run<5>(src, stamps, sink, cus, steps, 0, 0, 8, 4);
        run<6>(src, stamps, sink, cus, steps, 0, 0, 1, 1);   // rotated piece order per workgroup
        run<6>(src, stamps, sink, cus, steps, 0, 0, 8, 4);
        run<7>(src, stamps, sink, cus, steps, 0, 0, 1, 1);
        run<7>(src, stamps, sink, cus, steps, 0, 0, 8, 4);
        run<0>(src, stamps, sink, cus, steps, 0, 0, 4, 2);
        run<0>(src, stamps, sink, cus, steps, 0, 0, 8, 4);
        run<0>(src, stamps, sink, cus, steps, 0, 0, 16, 2);
        run<0>(src, stamps, sink, cus, steps, 0, 0, 32, 1);
        run<0>(src, stamps, sink, cus, steps, 0, 0, 32, 32);
    }
    hipFree(sink); hipFree(src); hipFree(stamps);
    return 0;
}
