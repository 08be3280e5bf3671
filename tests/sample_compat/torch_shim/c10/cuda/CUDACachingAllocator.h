/* TEST-HARNESS FIXTURE (see ../../ATen/cuda/CUDAContext.h): einsum.cc:22 includes this name; torch-ROCm has
 * the same allocator under c10/hip. */
#pragma once
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
