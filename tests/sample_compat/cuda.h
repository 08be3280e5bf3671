/* TEST-HARNESS FIXTURE — einsum.cc:24 includes <cuda.h> (driver API) without using any of it. */
#ifndef SAMPLE_COMPAT_CUDA_H_
#define SAMPLE_COMPAT_CUDA_H_
#include "cuda_runtime.h"
#endif
