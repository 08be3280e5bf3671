// Microbenchmark (tools only): A[a,b,c] -> C[c,a,b] fp32 at 2048^3 (BASELINE configs[2], elementwise_permute.cu:198-200 shape
// class) as a batched 2-D transpose a <-> c, swept over what elementwise.hip fixes today:
//   T0 x T1   tile extents along c (the output's contiguous mode: T0*4 bytes per written row segment) and along a (the
//             input's contiguous mode: T1*4 bytes per read row segment)
//   ORDER     which coordinate consecutive workgroup ids walk first:
//               0 = c-tile, a-tile, b   (elementwise.hip today: every c — 2048 distinct 16-MiB regions of A — live at once)
//               1 = a-tile, b, c-tile   (few c-tiles live at a time: small set of A regions, D rows written T0*4 B at a time)
//               2 = b, a-tile, c-tile
//               3 = a-tile, c-tile, b   (all of one b plane, a fastest)
//   XCD       1 = workgroup id remapped so that each XCD (id % 8) owns a contiguous run of the order above
//   NT        nontemporal loads/stores vs plain
// One JSON line per point: GB/s = 2 * 4 * 2048^3 / time (elementwise_permute.cu:208), verified against a direct gather.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/transpose_sweep.hip -o tools/ubench/transpose_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Shape { uint32_t Ea, Eb, Ec; };

// 256 threads.  Read: lanes along a (float4), rows along c.  LDS tile [a][c] (+4 pad).  Write: lanes along c, rows along a.
template <int T0, int T1, int ORDER, int XCD, int NT, int REV = 0>
__global__ void __launch_bounds__(256) tr(const float* __restrict__ A, float* __restrict__ D, Shape s, uint32_t nTiles) {
    constexpr int LD = T0 + 4;
    __shared__ __attribute__((aligned(16))) float tile[T1 * LD];
    const int tid = threadIdx.x;
    const uint32_t tc = s.Ec / T0, ta = s.Ea / T1;
    for (uint32_t blk = blockIdx.x; blk < nTiles; blk += gridDim.x) {
        uint32_t id = blk;
        if (XCD) {   // XCD x = id % 8 owns ids [x * nTiles/8, (x+1) * nTiles/8)
            const uint32_t per = nTiles / 8;
            id = (blk & 7) * per + (blk >> 3);
        }
        uint32_t ic, ia, ib;
        if (ORDER == 0)      { ic = id % tc; ia = (id / tc) % ta; ib = id / (tc * ta); }
        else if (ORDER == 1) { ia = id % ta; ib = (id / ta) % s.Eb; ic = id / (ta * s.Eb); }
        else if (ORDER == 2) { ib = id % s.Eb; ia = (id / s.Eb) % ta; ic = id / (s.Eb * ta); }
        else                 { ia = id % ta; ic = (id / ta) % tc; ib = id / (tc * ta); }
        const float* src = A + (size_t)ia * T1 + (size_t)ib * s.Ea + (size_t)ic * T0 * s.Ea * s.Eb;
        const size_t sDa = REV ? (size_t)s.Ec * s.Eb : (size_t)s.Ec, sDb = REV ? (size_t)s.Ec : (size_t)s.Ec * s.Ea;
        float*       dst = D + (size_t)ic * T0 + (size_t)ia * T1 * sDa + (size_t)ib * sDb;
        // ---- read T0 rows (c) x T1 floats (a): lanes cover T1/4 float4 per row, 256/(T1/4) rows per pass, 4 rows per lane group
        constexpr int LPR = T1 / 4;            // lanes per row
        constexpr int RPP = 256 / LPR;         // rows per pass (one row per lane)
        // each lane takes 4 consecutive c rows (register 4x4 transpose), so a pass covers 4*RPP rows
        constexpr int PASSES = T0 / (4 * RPP);
        static_assert(PASSES >= 1, "tile too small for 256 lanes");
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int la = 4 * (tid % LPR);                    // a offset
            const int lc = 4 * (tid / LPR) + ps * 4 * RPP;     // c offset (4 rows)
            f32x4 in[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4* ptr = reinterpret_cast<const f32x4*>(src + (size_t)(lc + r) * s.Ea * s.Eb + la);
                in[r] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 o = {in[0][j], in[1][j], in[2][j], in[3][j]};
                *reinterpret_cast<f32x4*>(&tile[(la + j) * LD + lc]) = o;
            }
        }
        __syncthreads();
        // ---- write T1 rows (a) x T0 floats (c)
        constexpr int LPW = T0 / 4;
        constexpr int RPW = 256 / LPW;
        constexpr int WP = T1 / RPW;
        static_assert(WP >= 1, "tile too small");
#pragma unroll
        for (int ps = 0; ps < WP; ++ps) {
            const int lc = 4 * (tid % LPW);
            const int la = tid / LPW + ps * RPW;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[la * LD + lc]);
            f32x4* ptr = reinterpret_cast<f32x4*>(dst + (size_t)la * sDa + lc);
            if (NT) __builtin_nontemporal_store(v, ptr); else *ptr = v;
        }
        __syncthreads();
    }
}

__global__ void fill(float* A, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32);
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        A[i] = (float)(x >> 8) * (1.0f / 16777216.0f);
    }
}
template <int REV>
__global__ void check(const float* A, const float* D, Shape s, unsigned long long* bad) {
    // 2^20 sampled positions
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    const uint32_t a = (uint32_t)(h >> 11) % s.Ea, b = (uint32_t)(h >> 29) % s.Eb, c = (uint32_t)(h >> 47) % s.Ec;
    const float x = A[(size_t)a + (size_t)b * s.Ea + (size_t)c * s.Ea * s.Eb];
    const float y = REV ? D[(size_t)c + (size_t)b * s.Ec + (size_t)a * s.Ec * s.Eb] : D[(size_t)c + (size_t)a * s.Ec + (size_t)b * s.Ec * s.Ea];
    if (x != y) atomicAdd(bad, 1ull);
}

static float* gA; static float* gD; static unsigned long long* gBad; static Shape gS;

template <int T0, int T1, int ORDER, int XCD, int NT, int REV = 0>
void point(unsigned gridCap) {
    const uint32_t nTiles = (gS.Ec / T0) * (gS.Ea / T1) * gS.Eb;
    const unsigned grid = gridCap && nTiles > gridCap ? gridCap : nTiles;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(gD, 0, 4096);
    tr<T0, T1, ORDER, XCD, NT, REV><<<grid, 256>>>(gA, gD, gS, nTiles);
    hipMemset(gBad, 0, 8);
    check<REV><<<4096, 256>>>(gA, gD, gS, gBad);
    unsigned long long bad = 0; hipMemcpy(&bad, gBad, 8, hipMemcpyDeviceToHost);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        tr<T0, T1, ORDER, XCD, NT, REV><<<grid, 256>>>(gA, gD, gS, nTiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = 2.0 * 4.0 * (double)gS.Ea * gS.Eb * gS.Ec;
    printf("{\"rev\": %d, \"T0\": %d, \"T1\": %d, \"order\": %d, \"xcd\": %d, \"nt\": %d, \"grid\": %u, \"ms\": %.3f, \"GBps\": %.1f, \"mismatches\": %llu}\n",
           REV, T0, T1, ORDER, XCD, NT, grid, best, bytes / (best * 1e-3) / 1e9, bad);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int T0, int T1>
void orders(unsigned cap) {
    point<T0, T1, 0, 0, 1>(cap);
    point<T0, T1, 0, 1, 1>(cap);
    point<T0, T1, 1, 0, 1>(cap);
    point<T0, T1, 1, 1, 1>(cap);
    point<T0, T1, 2, 0, 1>(cap);
    point<T0, T1, 2, 1, 1>(cap);
    point<T0, T1, 3, 0, 1>(cap);
    point<T0, T1, 3, 1, 1>(cap);
}

int main(int argc, char** argv) {
    const uint32_t E = argc > 1 ? (uint32_t)atoi(argv[1]) : 2048;
    gS = Shape{E, E, E};
    const size_t n = (size_t)E * E * E;
    if (hipMalloc(&gA, n * 4) != hipSuccess || hipMalloc(&gD, n * 4) != hipSuccess || hipMalloc(&gBad, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    fill<<<8192, 256>>>(gA, n);
    hipDeviceSynchronize();
    const unsigned cap = 256u * 8u * 16u;
    if (argc > 2 && atoi(argv[2]) == 3) {   // third sweep: the full reversal C[c,b,a] (both sides walk a 16-MiB pitch), one workgroup per tile
        point<64, 64, 0, 0, 1, 1>(0);
        point<128, 64, 0, 0, 1, 1>(0);
        point<256, 64, 0, 0, 1, 1>(0);
        point<256, 64, 0, 1, 1, 1>(0);
        point<256, 64, 1, 0, 1, 1>(0);
        point<256, 64, 1, 1, 1, 1>(0);
        point<256, 64, 2, 0, 1, 1>(0);
        point<256, 64, 2, 1, 1, 1>(0);
        point<256, 64, 3, 0, 1, 1>(0);
        point<256, 64, 3, 1, 1, 1>(0);
        point<128, 128, 0, 0, 1, 1>(0);
        point<128, 128, 1, 0, 1, 1>(0);
        point<128, 128, 2, 0, 1, 1>(0);
        point<128, 128, 3, 0, 1, 1>(0);
        point<128, 128, 3, 1, 1, 1>(0);
        point<256, 128, 0, 0, 1, 1>(0);
        point<64, 256, 0, 0, 1, 1>(0);
        point<256, 64, 0, 0, 1, 0>(0);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 2) {   // second sweep: one workgroup per tile (no grid-stride loop) vs capped grids
        point<64, 64, 0, 0, 1>(0);
        point<128, 32, 0, 0, 1>(0);
        point<128, 64, 0, 0, 1>(0);
        point<128, 64, 2, 0, 1>(0);
        point<128, 128, 0, 0, 1>(0);
        point<256, 32, 0, 0, 1>(0);
        point<256, 64, 0, 0, 1>(0);
        point<256, 64, 2, 0, 1>(0);
        point<256, 64, 3, 1, 1>(0);
        point<512, 32, 0, 0, 1>(0);
        point<512, 64, 0, 0, 1>(0);
        point<256, 64, 0, 0, 1>(cap * 4);
        point<256, 64, 0, 0, 1>(cap * 16);
        point<128, 64, 0, 0, 1>(cap * 4);
        point<128, 64, 0, 0, 1>(cap * 16);
        point<256, 64, 0, 0, 0>(0);
        point<128, 64, 0, 0, 0>(0);
        return 0;
    }
    orders<64, 64>(cap);
    orders<128, 64>(cap);
    orders<64, 128>(cap);
    orders<128, 128>(cap);
    orders<256, 64>(cap);
    orders<64, 256>(cap);
    // grid shape and cache policy on the base tile and the 128 x 128 tile
    point<64, 64, 0, 0, 0>(cap);
    point<64, 64, 0, 0, 1>(0);
    point<64, 64, 0, 0, 1>(256u * 8u);
    point<128, 128, 0, 0, 0>(cap);
    point<128, 128, 0, 0, 1>(0);
    point<128, 128, 1, 0, 1>(0);
    point<128, 128, 0, 0, 1>(256u * 2u);
    point<128, 128, 1, 0, 1>(256u * 2u);
    return 0;
}
