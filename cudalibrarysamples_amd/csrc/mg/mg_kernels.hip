// Device code of libcutensorMg.so (gfx950): the batched cell copy of the gather step.
//
// A device receives the cells it owns itself by device copies into its [cell][cell buffer] staging images (mg.cpp, step 1).  With a
// block-cyclic distribution a device owns several cells of each tensor (contraction_multi_gpu.cu:154-193: 2 x 2 grids of cells over the
// handle's devices), and hipMemcpyAsync costs the calling thread 2-3 us per cell — at eight handle devices the copies were the largest
// part of the host time of a cutensorMgContraction call (tools/mg_host_cost_n.py).  One launch moves up to kMgCopyBatch cells: the
// (source, destination, bytes) triples travel in the kernel arguments, blockIdx.y picks the cell, blockIdx.x walks it in 16-byte
// units (4 KiB per workgroup and step, grid-stride), so a batch streams at the rate of a flat device copy.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "mg_kernels.h"

namespace {

__global__ __launch_bounds__(256) void mg_copy_cells_kernel(MgCopyBatch b) {
    const int c = (int)blockIdx.y;
    const char* __restrict__ s = static_cast<const char*>(b.src[c]);
    char* __restrict__ d = static_cast<char*>(b.dst[c]);
    const uint64_t bytes = b.bytes[c];
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t nth = (uint64_t)gridDim.x * 256u;
    if ((((uintptr_t)s | (uintptr_t)d | bytes) & 15u) == 0) {
        const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(s);
        uint4* __restrict__ d4 = reinterpret_cast<uint4*>(d);
        const uint64_t n = bytes >> 4;
        uint64_t i = tid;
        // four independent 16-byte loads in flight per lane
        for (; i + 3 * nth < n; i += 4 * nth) {
            const uint4 v0 = s4[i], v1 = s4[i + nth], v2 = s4[i + 2 * nth], v3 = s4[i + 3 * nth];
            d4[i] = v0; d4[i + nth] = v1; d4[i + 2 * nth] = v2; d4[i + 3 * nth] = v3;
        }
        for (; i < n; i += nth) d4[i] = s4[i];
    } else if ((((uintptr_t)s | (uintptr_t)d | bytes) & 3u) == 0) {
        const uint32_t* __restrict__ s1 = reinterpret_cast<const uint32_t*>(s);
        uint32_t* __restrict__ d1 = reinterpret_cast<uint32_t*>(d);
        const uint64_t n = bytes >> 2;
        for (uint64_t i = tid; i < n; i += nth) d1[i] = s1[i];
    } else {
        for (uint64_t i = tid; i < bytes; i += nth) d[i] = s[i];
    }
}

}  // namespace

hipError_t mg_copy_cells(const MgCopyBatch& batch, hipStream_t stream) {
    if (batch.n <= 0) return hipSuccess;
    uint64_t most = 0;
    for (int i = 0; i < batch.n; ++i) most = batch.bytes[i] > most ? batch.bytes[i] : most;
    if (most == 0) return hipSuccess;
    // 16 KiB per workgroup and sweep; enough workgroups over the batch to cover the chip a few times, no more
    uint64_t gx = (most + 16383) / 16384;
    const uint64_t cap = (uint64_t)(4096 / batch.n) > 64 ? (uint64_t)(4096 / batch.n) : 64;
    if (gx > cap) gx = cap;
    if (gx == 0) gx = 1;
    mg_copy_cells_kernel<<<dim3((unsigned)gx, (unsigned)batch.n), dim3(256), 0, stream>>>(batch);
    return hipGetLastError();
}
