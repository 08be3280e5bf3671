// Microbenchmark (tools only): is the 6.9 us a workgroup of gett_h16w4x_kernel needs to get its 256 x 256 bf16 tile (128 KiB) out a
// property of the CU's own store path or of 256 CUs storing at once?  G workgroups (one per CU, 4 waves), every wave stores 32 KiB as
// 32 nontemporal 16-byte-per-lane instructions laid out as the kernel's epilogue does (4 rows x 256 B per instruction, row pitch
// 16 KiB); reported: shader cycles from the first store to (a) the last store ISSUED and (b) all stores complete (s_waitcnt vmcnt(0)),
// mean over the workgroups, for G = 8 .. 256.   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/store_rate_by_cus.hip -o tools/ubench/store_rate_by_cus
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short s16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256, 1) k(uint16_t* D, unsigned long long* stamps, size_t pitch) {
    __shared__ char pad[96 * 1024];          // one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 9999) pad[threadIdx.x] = 1;
    const int tm = blockIdx.x >> 5, tn = blockIdx.x & 31;
    uint16_t* T = D + (size_t)tm * 256 * pitch + (size_t)tn * 256 + (size_t)(wave >> 1) * 128 * pitch + (size_t)(wave & 1) * 128;
    s16x8 v = {(short)lane, (short)wave, 3, 4, 5, 6, 7, 8};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 32; ++i) {           // rows 4 i + (lane >> 4) of the wave's 128 x 128 quadrant, 16 lanes x 16 B = 256 B each
        s16x8* dst = reinterpret_cast<s16x8*>(T + (size_t)(4 * i + (lane >> 4)) * pitch + (lane & 15) * 8);
        __builtin_nontemporal_store(v, dst);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = t1 - t0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

int main() {
    uint16_t* D; hipMalloc(&D, (size_t)8192 * 8192 * 2);
    hipMemset(D, 0, (size_t)8192 * 8192 * 2);
    unsigned long long* stamps; hipMalloc(&stamps, 1024 * 4 * 2 * 8);
    for (int G : {8, 32, 64, 128, 256, 512, 1024}) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, nullptr, D, stamps, (size_t)8192);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)G * 8);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        double issued = 0, done = 0;
        for (int i = 0; i < G * 4; ++i) { issued += (double)h[2 * i]; done += (double)h[2 * i + 1]; }
        issued /= G * 4; done /= G * 4;
        printf("{\"workgroups\": %d, \"cycles_until_issued\": %.0f, \"cycles_until_complete\": %.0f, \"bytes_per_clk_per_cu_issue\": %.1f, \"bytes_per_clk_per_cu_complete\": %.1f}\n",
               G, issued, done, 131072.0 / issued, 131072.0 / done);
    }
    return 0;
}
