// Microbenchmark (tools only, round 6): what the fp32 GETT epilogue's store pattern costs a CU.  A workgroup (4 waves) stores a 128 x 128
// fp32 tile (64 KiB) of a column-major M x N output (m contiguous, pitch M floats), each wave its 64 x 64 quadrant as 16 instructions of
// 16 bytes per lane:
//   pattern 0: accumulator order, as gett_store_tile_f32 does — per instruction 16 columns x 64 B (a lane's four registers of a 16 x 16
//              fragment are four consecutive m; lane & 15 = column), fragments i (m) outer, j (n) inner
//   pattern 1: whole rows — per instruction 4 columns x 256 B (lane & 15 = 16-byte unit along m, lane >> 4 = column), what a transposition
//              through LDS would give
// each nontemporal and plain.  G workgroups (one per CU); shader cycles from the first store to the last one ISSUED and to all COMPLETE.
// hipcc --offload-arch=gfx950 -O3 -w tools/ubench/store_f32_pieces.hip -o tools/ubench/store_f32_pieces
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool NT>
__global__ void __launch_bounds__(256, 2) k(float* D, unsigned long long* stamps, size_t pitch, int tilesM) {
    __shared__ char pad[64 * 1024];          // two workgroups per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 9999) pad[threadIdx.x] = 1;
    const int tm = blockIdx.x % tilesM, tn = blockIdx.x / tilesM;
    float* T = D + (size_t)tn * 128 * pitch + (size_t)tm * 128 + (size_t)(wave >> 1) * 64 * pitch + (size_t)(wave & 1) * 64;
    f32x4 v = {(float)lane, (float)wave, 3.f, 4.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4* dst;
            if (PAT == 0) dst = reinterpret_cast<f32x4*>(T + (size_t)(16 * j + (lane & 15)) * pitch + 16 * i + 4 * (lane >> 4));
            else          dst = reinterpret_cast<f32x4*>(T + (size_t)(16 * i + 4 * j + (lane >> 4)) * pitch + 4 * (lane & 15));
            if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (lane == 0) { stamps[(blockIdx.x * 4 + wave) * 2] = t1 - t0; stamps[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

template <int PAT, bool NT>
static void run(float* D, unsigned long long* stamps, size_t M) {
    for (int G : {8, 256, 1024, 16384}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<PAT, NT>), dim3(G), dim3(256), 0, nullptr, D, stamps, M, (int)(M / 128));
        hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL((k<PAT, NT>), dim3(G), dim3(256), 0, nullptr, D, stamps, M, (int)(M / 128));
        hipEventRecord(e1, nullptr);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)G * 8);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        double issued = 0, done = 0;
        for (int i = 0; i < G * 4; ++i) { issued += (double)h[2 * i]; done += (double)h[2 * i + 1]; }
        issued /= G * 4; done /= G * 4;
        printf("{\"pattern\": %d, \"nontemporal\": %d, \"pitch_floats\": %zu, \"workgroups\": %d, \"cycles_until_issued\": %.0f, \"cycles_until_complete\": %.0f, "
               "\"bytes_per_clk_per_cu_issue\": %.1f, \"bytes_per_clk_per_cu_complete\": %.1f, \"launch_us\": %.1f, \"TBps\": %.2f}\n",
               PAT, (int)NT, M, G, issued, done, 65536.0 / issued, 65536.0 / done, ms * 1e3, (double)G * 65536.0 / (ms * 1e-3) / 1e12);
    }
}

int main() {
    const size_t M = 16384, N = 16384;
    float* D; hipMalloc(&D, M * N * 4);
    hipMemset(D, 0, M * N * 4);
    unsigned long long* stamps; hipMalloc(&stamps, 16384 * 4 * 2 * 8);
    run<0, true>(D, stamps, M); run<0, false>(D, stamps, M); run<1, true>(D, stamps, M); run<1, false>(D, stamps, M);
    return 0;
}
