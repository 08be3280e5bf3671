// Device helpers of libcutensorMg.so (mg_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int kMgCopyBatch = 16;    // cells per launch: 16 x (source, destination, bytes) = 388 bytes of kernel arguments

struct MgCopyBatch {
    const void* src[kMgCopyBatch];
    void* dst[kMgCopyBatch];
    uint64_t bytes[kMgCopyBatch];
    int n = 0;
};

// dst[i][0 .. bytes[i]) = src[i][0 .. bytes[i]) for i < n, one launch on `stream` of the current device (same-device memory)
hipError_t mg_copy_cells(const MgCopyBatch& batch, hipStream_t stream);
