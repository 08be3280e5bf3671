// Microbenchmark (tools only): what does buffer_load_dword{,x4} ... lds do when a lane's global address is NOT 16-byte aligned?
// The aligned 16-bit / fp32 GETT kernels stage operands with 16-byte LDS-DMA units and the planner therefore demands that every row
// pitch is a multiple of 16 bytes (plan_contraction.cpp, LAY_K / LAY_F).  A row pitch of 8200 bytes (bf16, extent 4100) puts every
// second row at 8 (mod 16), 4098 at 4 (mod 16), 4097 at 2 (mod 16).  Three questions:
//   1. correctness — does the LDS piece hold the 16 bytes at the misaligned address, for every misalignment 0, 2, .., 14?
//   2. range check — a unit that straddles the end of the buffer (num_records): which of its dwords arrive, which are zero?
//   3. rate — B/clk/CU of the K-contiguous staging pattern (a piece = 8 rows x 128 B) at pitch 8192 / 8200 / 8196 / 8194 out of L2.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/ldsdma_unaligned.hip -o tools/ubench/ldsdma_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int rsrc_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(uint64_t addr, uint32_t records) {
    rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)addr);
    r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(addr >> 32));
    r[2] = __builtin_amdgcn_readfirstlane((int)records);
    r[3] = 0x00020000;
    return r;
}

// MODE 0: dwordx4 (16 B per lane, LDS piece of 1 KiB); 1: dword (4 B per lane, 256 B); 2: ushort (lane writes a dword holding the
// zero-extended 16 bits, 256 B)
template <int MODE>
__global__ void __launch_bounds__(64, 1) land(const char* src, uint32_t records, uint32_t pitch, uint32_t mis, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) char lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const rsrc_t rs = make_rsrc((uint64_t)(uintptr_t)src, records);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t off = (uint32_t)lane * pitch + mis;
    if constexpr (MODE == 0)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(ldsBase), "v"(off), "s"(rs) : "memory");
    else if constexpr (MODE == 1)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(ldsBase), "v"(off), "s"(rs) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_ushort %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(ldsBase), "v"(off), "s"(rs) : "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}

template <int MODE>
static void check_land(const char* dsrc, const std::vector<uint8_t>& hsrc, uint32_t* dout, uint32_t pitch) {
    const int bytesPerLane = MODE == 0 ? 16 : 4;
    for (uint32_t mis = 0; mis < 16; mis += 2) {
        hipLaunchKernelGGL((land<MODE>), dim3(1), dim3(64), 0, nullptr, dsrc, 0xffffffffu, pitch, mis, dout);
        hipDeviceSynchronize();
        std::vector<uint8_t> got(1024);
        hipMemcpy(got.data(), dout, 1024, hipMemcpyDeviceToHost);
        int bad = 0, firstBad = -1;
        for (int lane = 0; lane < 64; ++lane)
            for (int b = 0; b < bytesPerLane; ++b) {
                uint8_t want = hsrc[(size_t)lane * pitch + mis + b];
                if (MODE == 2 && b >= 2) want = 0;
                if (got[lane * bytesPerLane + b] != want) { ++bad; if (firstBad < 0) firstBad = lane * bytesPerLane + b; }
            }
        printf("{\"test\":\"land\",\"op\":\"%s\",\"pitch\":%u,\"misalign\":%u,\"wrong_bytes\":%d,\"first_wrong\":%d}\n",
               MODE == 0 ? "dwordx4" : MODE == 1 ? "dword" : "ushort", pitch, mis, bad, firstBad);
    }
}

// 16-byte global stores / loads at a misaligned address (the epilogue's D rows at an 8200- or 8194-byte pitch)
__global__ void __launch_bounds__(64, 1) store16(char* dst, const char* src, uint32_t pitch, uint32_t mis) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x;
    const i32x4 v = *reinterpret_cast<const i32x4 __attribute__((address_space(1)))*>((uintptr_t)(src + (size_t)lane * pitch + mis));
    __builtin_nontemporal_store(v, reinterpret_cast<i32x4 __attribute__((address_space(1)))*>((uintptr_t)(dst + (size_t)lane * pitch + mis)));
}
static void check_store(char* ddst, const char* dsrc, const std::vector<uint8_t>& hsrc, size_t n) {
    for (uint32_t pitch : {64u, 8200u, 8194u})
        for (uint32_t mis = 0; mis < 16; mis += 2) {
            hipMemset(ddst, 0, n);
            hipLaunchKernelGGL(store16, dim3(1), dim3(64), 0, nullptr, ddst, dsrc, pitch, mis);
            hipDeviceSynchronize();
            std::vector<uint8_t> got(n);
            hipMemcpy(got.data(), ddst, n, hipMemcpyDeviceToHost);
            std::vector<uint8_t> want(n, 0);
            for (int lane = 0; lane < 64; ++lane)
                for (int b = 0; b < 16; ++b) want[(size_t)lane * pitch + mis + b] = hsrc[(size_t)lane * pitch + mis + b];
            size_t bad = 0;
            for (size_t i = 0; i < n; ++i) bad += got[i] != want[i];
            printf("{\"test\":\"global load + nontemporal store of 16 B\",\"pitch\":%u,\"misalign\":%u,\"wrong_bytes\":%zu}\n", pitch, mis, bad);
        }
}

// range check WITH a scalar offset (the fp32 streaming kernels put the K-tile's byte offset into soffset): is soffset part of what is
// compared with num_records?  Lane l reads 16 bytes at soff + (records - soff - 16 + 4 l).
__global__ void __launch_bounds__(64, 1) land_soff(const char* src, uint32_t records, uint32_t soff, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) char lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const rsrc_t rs = make_rsrc((uint64_t)(uintptr_t)src, records);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t off = records - soff - 16u + 4u * (uint32_t)lane;
    const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)soff);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(ldsBase), "v"(off), "s"(rs), "s"(so) : "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}
static void check_range_soff(const char* dsrc, const std::vector<uint8_t>& hsrc, uint32_t* dout) {
    const uint32_t records = 8192u;
    for (uint32_t soff : {0u, 4096u}) {
        hipLaunchKernelGGL(land_soff, dim3(1), dim3(64), 0, nullptr, dsrc, records, soff, dout);
        hipDeviceSynchronize();
        std::vector<uint8_t> got(1024);
        hipMemcpy(got.data(), dout, 1024, hipMemcpyDeviceToHost);
        for (int lane = 0; lane <= 5; ++lane) {
            char line[64] = {0};
            for (int b = 0; b < 16; ++b) {
                const size_t at = (size_t)records - 16 + 4 * lane + b;
                const uint8_t g = got[lane * 16 + b];
                line[b] = (g == hsrc[at]) ? (at < records ? 'd' : 'X') : (g == 0 ? '0' : '?');
            }
            printf("{\"test\":\"range with soffset\",\"records\":%u,\"soffset\":%u,\"unit_start_rel_end\":%d,\"bytes\":\"%s\"}\n", records, soff, 4 * lane - 16, line);
        }
    }
}

// range check: lanes 0..8 read 16 bytes at records - 16 + 2 * lane (lane 0 whole inside, lane 8 whole outside)
static void check_range(const char* dsrc, const std::vector<uint8_t>& hsrc, uint32_t* dout) {
    for (uint32_t records : {4096u, 4098u, 4100u, 4104u}) {
        // pitch 2: lane l reads at mis + 2 l
        hipLaunchKernelGGL((land<0>), dim3(1), dim3(64), 0, nullptr, dsrc, records, 2u, records - 16u, dout);
        hipDeviceSynchronize();
        std::vector<uint8_t> got(1024);
        hipMemcpy(got.data(), dout, 1024, hipMemcpyDeviceToHost);
        for (int lane = 0; lane <= 9; ++lane) {
            char line[64] = {0};
            for (int b = 0; b < 16; ++b) {
                const size_t at = (size_t)records - 16 + 2 * lane + b;
                const uint8_t g = got[lane * 16 + b];
                line[b] = (g == hsrc[at]) ? (at < records ? 'd' : 'X') : (g == 0 ? '0' : '?');   // d = data (in range), X = data from BEYOND the range, 0 = zero
            }
            printf("{\"test\":\"range\",\"records\":%u,\"lane_start\":%d,\"bytes\":\"%s\"}\n", records, (int)(2 * lane) - 16, line);
        }
    }
}

// K-contiguous staging pattern: a piece = 8 rows x 8 units of 16 B (one 128-byte row segment per row), rows `pitch` bytes apart.
template <int W, int Q>
__global__ void __launch_bounds__(64 * W, 1) stream(const char* __restrict__ src, unsigned long long* stamps, uint32_t pitch, uint32_t share, int pieces) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t footprint = 256u * pitch;
    const uint32_t region = share ? (blockIdx.x & 7u) : blockIdx.x;
    const rsrc_t rs = make_rsrc((uint64_t)(uintptr_t)(src + (size_t)region * (footprint + 4096u)), footprint);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t kTiles = pitch / 128u;                       // whole 128-byte segments of a row
    const uint32_t rowInPiece = (uint32_t)(lane >> 3), unit = (uint32_t)(lane & 7);
    uint32_t kt = share ? (blockIdx.x >> 3) % kTiles : 0u, rb = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < pieces; ++i) {
        const uint32_t row = (rb * (8u * W) + (uint32_t)wave * 8u + rowInPiece) & 255u;
        const uint32_t off = row * pitch + kt * 128u + unit * 16u;
        const uint32_t slot = __builtin_amdgcn_readfirstlane(ldsBase + (uint32_t)((wave * 8 + (i & 7)) * 1024) % (64u * 1024u));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(slot), "v"(off), "s"(rs) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q - 1) : "memory");
        rb += 1u;
        if (rb == 256u / (8u * W)) { rb = 0; kt = (kt + 1u == kTiles) ? 0u : kt + 1u; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

// free-contiguous staging pattern: a piece = 4 k-rows x 16 units (256 contiguous bytes per k-row), k-rows `pitch` bytes apart
template <int W, int Q>
__global__ void __launch_bounds__(64 * W, 1) streamF(const char* __restrict__ src, unsigned long long* stamps, uint32_t pitch, uint32_t share, int pieces) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t footprint = 256u * pitch;
    const uint32_t region = share ? (blockIdx.x & 7u) : blockIdx.x;
    const rsrc_t rs = make_rsrc((uint64_t)(uintptr_t)(src + (size_t)region * (footprint + 4096u)), footprint);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t segs = pitch / 256u;
    const uint32_t kInPiece = (uint32_t)(lane >> 4), unit = (uint32_t)(lane & 15);
    uint32_t seg = share ? (blockIdx.x >> 3) % segs : 0u, kb = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < pieces; ++i) {
        const uint32_t krow = (kb * (4u * W) + (uint32_t)wave * 4u + kInPiece) & 255u;
        const uint32_t off = krow * pitch + seg * 256u + unit * 16u;
        const uint32_t slot = __builtin_amdgcn_readfirstlane(ldsBase + (uint32_t)((wave * 8 + (i & 7)) * 1024) % (64u * 1024u));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(slot), "v"(off), "s"(rs) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q - 1) : "memory");
        kb += 1u;
        if (kb == 256u / (4u * W)) { kb = 0; seg = (seg + 1u == segs) ? 0u : seg + 1u; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

template <int W, int Q, bool F>
static void run(const char* src, unsigned long long* stamps, int cus, uint32_t pitch, uint32_t share) {
    const int pieces = 8192;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&] {
        if (F) hipLaunchKernelGGL((streamF<W, Q>), dim3(cus), dim3(64 * W), 0, nullptr, src, stamps, pitch, share, pieces);
        else   hipLaunchKernelGGL((stream<W, Q>), dim3(cus), dim3(64 * W), 0, nullptr, src, stamps, pitch, share, pieces);
    };
    go(); go();
    hipEventRecord(e0, nullptr);
    go();
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
    printf("{\"test\":\"rate\",\"pattern\":\"%s\",\"pitch\":%u,\"pitch_mod_16\":%u,\"source\":\"%s\",\"waves\":%d,\"in_flight\":%d,\"bytes_per_clk_per_cu\":%.1f,\"TBps_chip\":%.2f}\n",
           F ? "free-contiguous (4 k-rows x 256 B)" : "K-contiguous (8 rows x 128 B)", pitch, pitch % 16u,
           share ? "panel shared by the 32 workgroups of an XCD (L2)" : "panel per workgroup (HBM / Infinity Cache)", W, Q, bytesWg / cyc,
           bytesWg * cus / (ms * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    // ---- correctness -------------------------------------------------------------------------------------------------
    const size_t n = 1u << 20;
    std::vector<uint8_t> h(n);
    uint32_t s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint8_t)(s >> 24) | 1u; }   // never zero
    char* d = nullptr;
    uint32_t* out = nullptr;
    hipMalloc((void**)&d, n);
    hipMalloc((void**)&out, 1024);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    check_land<0>(d, h, out, 64u);
    check_land<0>(d, h, out, 8200u);
    check_land<0>(d, h, out, 8194u);
    check_land<1>(d, h, out, 64u);
    check_land<1>(d, h, out, 8194u);
    check_land<2>(d, h, out, 66u);
    check_range(d, h, out);
    check_range_soff(d, h, out);
    // descriptor BASE misaligned by 2 / 6 / 10 bytes (the kernels put a per-wave minimum offset into the base)
    for (uint32_t b : {2u, 6u, 10u}) {
        std::vector<uint8_t> hs(h.begin() + b, h.end());
        printf("{\"note\":\"descriptor base + %u bytes\"}\n", b);
        check_land<0>(d + b, hs, out, 8200u);
    }
    {
        char* dd = nullptr;
        hipMalloc((void**)&dd, n);
        check_store(dd, d, h, n);
        hipFree(dd);
    }
    hipFree(d); hipFree(out);
    // ---- rate ----------------------------------------------------------------------------------------------------------
    const size_t bytes = (size_t)cus * (256u * 8448u + 4096u);
    char* src = nullptr;
    unsigned long long* stamps = nullptr;
    hipMalloc((void**)&src, bytes);
    hipMemset(src, 1, bytes);
    hipMalloc((void**)&stamps, 2 * (size_t)cus * 8);
    for (uint32_t share : {1u, 0u})
        for (uint32_t pitch : {8192u, 8448u, 8200u, 8196u, 8194u, 8208u}) {
            run<4, 16, false>(src, stamps, cus, pitch, share);
            run<4, 16, true>(src, stamps, cus, pitch, share);
        }
    hipFree(src); hipFree(stamps);
    return 0;
}
