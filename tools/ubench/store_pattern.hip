// Microbenchmark (tools only): how fast does a workgroup get a 256 x 256 bf16 output tile (128 KiB) out, depending on how
// the 16-byte stores of a wave-instruction are laid over the rows of D (row pitch 16 KiB, the 8192^3 case)?  1024 workgroups
// of 8 waves, each wave stores 16 KiB with 16 instructions, 128 MiB in all; the store values come from registers.
//   pattern 0: 16 rows x 64 B per instruction  (the epilogue as it is: a wave owns 32-column fragments)
//   pattern 1:  8 rows x 128 B                 (whole cache lines)
//   pattern 2:  4 rows x 256 B
//   pattern 3:  2 rows x 512 B                 (whole tile rows: needs a workgroup-wide turn through LDS)
//   pattern 4:  1 KiB contiguous               (not a layout a tile of D can have; the upper bound)
// each with nontemporal and with plain stores.   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_pattern.hip -o tools/ubench/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int PAT, int NT>
__global__ void __launch_bounds__(512, 1) k(uint16_t* D, int tilesN, size_t pitch, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    uint16_t* T = D + (size_t)tm * 256 * pitch + (size_t)tn * 256;
    s16x8 v = {(short)lane, (short)wave, 3, 4, 5, 6, 7, 8};
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            size_t off;
            if (PAT == 0) {          // wave: 32-column strip (wave & 3) of halves..., rows 16 i' ...: 16 rows x 4 lanes
                const int strip = (wave & 3) * 32 + (i & 1) * 128, rowBase = (wave >> 2) * 128 + (i >> 1) * 16;
                off = (size_t)(rowBase + (lane >> 2)) * pitch + strip + (lane & 3) * 8;
            } else if (PAT == 1) {   // 8 rows x 8 lanes (128 B)
                const int strip = (wave & 3) * 64, rowBase = (wave >> 2) * 128 + i * 8;
                off = (size_t)(rowBase + (lane >> 3)) * pitch + strip + (lane & 7) * 8;
            } else if (PAT == 2) {   // 4 rows x 16 lanes (256 B)
                const int strip = (wave & 1) * 128, rowBase = (wave >> 1) * 64 + i * 4;
                off = (size_t)(rowBase + (lane >> 4)) * pitch + strip + (lane & 15) * 8;
            } else if (PAT == 3) {   // 2 rows x 32 lanes (512 B)
                const int rowBase = wave * 32 + i * 2;
                off = (size_t)(rowBase + (lane >> 5)) * pitch + (lane & 31) * 8;
            } else {                 // contiguous (ignores the tile shape)
                off = ((size_t)blockIdx.x * 8 + wave) * 8192 + (size_t)i * 512 + lane * 8;
            }
            s16x8* dst = reinterpret_cast<s16x8*>((PAT == 4 ? D : T) + off);
            if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
        v[2] += 1;
    }
}

template <int PAT, int NT>
void point(uint16_t* D) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<PAT, NT><<<1024, 512>>>(D, 32, 8192, 1);
    hipEventRecord(e0);
    const int launches = 20;
    for (int w = 0; w < launches; ++w) k<PAT, NT><<<1024, 512>>>(D, 32, 8192, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / launches;
    printf("{\"pattern\": %d, \"nontemporal\": %d, \"us_per_launch\": %.2f, \"TBps\": %.2f, \"us_per_workgroup_round\": %.2f}\n", PAT, NT, us,
           134217728.0 / (us * 1e-6) / 1e12, us / 4);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    uint16_t* D; hipMalloc(&D, (size_t)8192 * 8192 * 2);
    hipMemset(D, 0, (size_t)8192 * 8192 * 2);
    point<0, 1>(D); point<1, 1>(D); point<2, 1>(D); point<3, 1>(D); point<4, 1>(D);
    point<0, 0>(D); point<1, 0>(D); point<2, 0>(D); point<3, 0>(D); point<4, 0>(D);
    return 0;
}
