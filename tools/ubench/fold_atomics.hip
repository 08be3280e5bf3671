// Microbenchmark (tools only): what would the headline einsum's split-K fold cost if the 256 workgroups accumulated their
// 96 x 96 fp32 partial tiles (36 KB each) with fp32 atomics into ONE tile per XCD instead of writing 256 partial tiles
// (9.4 MB, write-through) that a second kernel folds?  No compute: every workgroup only delivers its tile, all at once — the
// end of the real kernel, where the 256 CUs finish within ~1 us of each other.
//   A  write-through 16-byte stores of 256 partial tiles + fold kernel (256-way sum)            — what the engine does
//   B  workgroup-scope atomic adds into tile[XCC_ID] + fold kernel (8-way sum, clears the tiles) — stays in the XCD's L2
//   C  agent-scope atomic adds into tile[XCC_ID] + the same fold                                  — memory-side atomics
//   D  agent-scope atomic adds into ONE tile + a 1-way "fold" (clear only)
// Reported: microseconds per (deliver + fold) pair, back to back in one stream, and of each kernel alone; sums are checked
// (every output must be exactly 256: the partial values are 1.0f).
// The atomic variants are NOT bit-reproducible for real data (the order of the 32 / 256 adds varies run to run).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fold_atomics.hip -o tools/ubench/fold_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int TILE = 96 * 96;          // floats
constexpr int WGS = 256, THREADS = 512;

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

__global__ void __launch_bounds__(THREADS) deliver_stores(float* partial) {
    // 9216 floats = 2304 quads; 512 lanes -> 4.5 quads per lane: lanes 0..255 store 5, the rest 4
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(partial + (size_t)blockIdx.x * TILE, 0, -1, 0x00020000);
    const f32x4 v = {1.f, 1.f, 1.f, 1.f};
    for (int q = threadIdx.x; q < TILE / 4; q += THREADS)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, q * 16, 0, /*sc1*/ 16);
}
__global__ void __launch_bounds__(256) fold_256(const float* partial, float* D) {
    // 8 quads per workgroup x 32 slice groups, 8 loads per lane in flight (the engine's splitk_reduce_frag_flat_kernel)
    __shared__ f32x4 red[4][8];
    const int q = threadIdx.x & 7, g = threadIdx.x >> 3;
    const uint32_t e = blockIdx.x * 8 + q;
    const f32x4* src = reinterpret_cast<const f32x4*>(partial) + e;
    f32x4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = __builtin_nontemporal_load(src + (size_t)(g + 32 * u) * (TILE / 4));
    f32x4 sum = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
#pragma unroll
    for (int m = 8; m < 64; m <<= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) sum[c] += __shfl_xor(sum[c], m, 64);
    if ((threadIdx.x & 63) < 8) red[threadIdx.x >> 6][q] = sum;
    __syncthreads();
    if (threadIdx.x >= 8) return;
    sum = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
    reinterpret_cast<f32x4*>(D)[e] = sum;
}

template <int SCOPE, bool PER_XCD>   // SCOPE 0 = workgroup, 1 = agent
__global__ void __launch_bounds__(THREADS) deliver_atomics(float* tiles, uint32_t* seen) {
    const uint32_t x = PER_XCD ? xcc_id() : 0u;
    if (threadIdx.x == 0 && seen != nullptr) atomicAdd(&seen[x], 1u);
    float* t = tiles + (size_t)x * TILE;
    for (int i = threadIdx.x; i < TILE; i += THREADS) {
        if (SCOPE == 0) (void)__hip_atomic_fetch_add(t + i, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else            (void)__hip_atomic_fetch_add(t + i, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int N>
__global__ void __launch_bounds__(256) fold_n_and_clear(float* tiles, float* D) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;          // one quad per lane
    if (e >= TILE / 4) return;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < N; ++x) {
        f32x4* p = reinterpret_cast<f32x4*>(tiles + (size_t)x * TILE) + e;
        sum += *p;
        *p = zero;                                              // self-clearing: ready for the next launch
    }
    reinterpret_cast<f32x4*>(D)[e] = sum;
}

static float time_us(void (*body)(), int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) body();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) body();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3f / iters;
}

static float *gPartial, *gTiles, *gD; static uint32_t* gSeen;
static void pairA() { deliver_stores<<<WGS, THREADS>>>(gPartial); fold_256<<<TILE / 4 / 8, 256>>>(gPartial, gD); }
static void onlyA1() { deliver_stores<<<WGS, THREADS>>>(gPartial); }
static void onlyA2() { fold_256<<<TILE / 4 / 8, 256>>>(gPartial, gD); }
static void pairB() { deliver_atomics<0, true><<<WGS, THREADS>>>(gTiles, nullptr); fold_n_and_clear<8><<<(TILE / 4 + 255) / 256, 256>>>(gTiles, gD); }
static void onlyB1() { deliver_atomics<0, true><<<WGS, THREADS>>>(gTiles, nullptr); }
static void pairC() { deliver_atomics<1, true><<<WGS, THREADS>>>(gTiles, nullptr); fold_n_and_clear<8><<<(TILE / 4 + 255) / 256, 256>>>(gTiles, gD); }
static void onlyC1() { deliver_atomics<1, true><<<WGS, THREADS>>>(gTiles, nullptr); }
static void pairD() { deliver_atomics<1, false><<<WGS, THREADS>>>(gTiles, nullptr); fold_n_and_clear<1><<<(TILE / 4 + 255) / 256, 256>>>(gTiles, gD); }
static void onlyF8() { fold_n_and_clear<8><<<(TILE / 4 + 255) / 256, 256>>>(gTiles, gD); }
static void empty2() { fold_n_and_clear<1><<<1, 64>>>(gTiles + 8 * TILE, gD + TILE); fold_n_and_clear<1><<<1, 64>>>(gTiles + 8 * TILE, gD + TILE); }

static int check(const char* what, float want) {
    std::vector<float> h(TILE);
    hipMemcpy(h.data(), gD, TILE * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (float v : h) bad += (v != want);
    if (bad) printf("# %s: %d of %d outputs differ from %.0f (first %.1f)\n", what, bad, TILE, want, h[0]);
    return bad;
}

int main() {
    hipMalloc(&gPartial, (size_t)WGS * TILE * 4); hipMalloc(&gTiles, (size_t)9 * TILE * 4); hipMalloc(&gD, 2 * TILE * 4); hipMalloc(&gSeen, 64);
    hipMemset(gTiles, 0, (size_t)9 * TILE * 4); hipMemset(gSeen, 0, 64);
    // where do 256 workgroups of 512 threads land?  (one per CU: 32 per XCD expected)
    deliver_atomics<1, true><<<WGS, THREADS>>>(gTiles, gSeen);
    fold_n_and_clear<8><<<(TILE / 4 + 255) / 256, 256>>>(gTiles, gD);
    uint32_t seen[16]; hipMemcpy(seen, gSeen, 64, hipMemcpyDeviceToHost);
    printf("{\"workgroups_per_xcc_id\": [%u,%u,%u,%u,%u,%u,%u,%u], \"others\": %u}\n", seen[0], seen[1], seen[2], seen[3], seen[4], seen[5], seen[6], seen[7],
           seen[8] + seen[9] + seen[10] + seen[11] + seen[12] + seen[13] + seen[14] + seen[15]);
    int bad = check("agent-scope warm-up", 256.f);
    const int iters = 2000;
    pairA(); hipDeviceSynchronize(); bad += check("A", 256.f);
    pairB(); hipDeviceSynchronize(); bad += check("B (workgroup-scope atomics per XCD)", 256.f);
    pairC(); hipDeviceSynchronize(); bad += check("C", 256.f);
    pairD(); hipDeviceSynchronize(); bad += check("D", 256.f);
    const float tE = time_us(empty2, iters);
    const float tA = time_us(pairA, iters), tA1 = time_us(onlyA1, iters), tA2 = time_us(onlyA2, iters);
    const float tB = time_us(pairB, iters), tB1 = time_us(onlyB1, iters);
    hipMemset(gTiles, 0, (size_t)9 * TILE * 4);
    const float tC = time_us(pairC, iters), tC1 = time_us(onlyC1, iters);
    hipMemset(gTiles, 0, (size_t)9 * TILE * 4);
    const float tD = time_us(pairD, iters);
    hipMemset(gTiles, 0, (size_t)9 * TILE * 4);
    const float tF8 = time_us(onlyF8, iters);
    printf("{\"us_two_empty_kernels\": %.2f, \"A_stores_plus_fold256\": %.2f, \"A_stores_alone\": %.2f, \"A_fold256_alone\": %.2f, "
           "\"B_wgscope_atomics_per_xcd_plus_fold8\": %.2f, \"B_atomics_alone\": %.2f, \"C_agent_atomics_per_xcd_plus_fold8\": %.2f, \"C_atomics_alone\": %.2f, "
           "\"D_agent_atomics_one_tile_plus_clear\": %.2f, \"fold8_alone\": %.2f, \"wrong_outputs\": %d}\n",
           tE, tA, tA1, tA2, tB, tB1, tC, tC1, tD, tF8, bad);
    return 0;
}
