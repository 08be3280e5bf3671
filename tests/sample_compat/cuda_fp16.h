/* TEST-HARNESS FIXTURE (see cuda_runtime.h in this directory): einsum.cu:28 includes <cuda_fp16.h>
 * for the __half type. */
#ifndef SAMPLE_COMPAT_CUDA_FP16_H_
#define SAMPLE_COMPAT_CUDA_FP16_H_
#include <hip/hip_fp16.h>
#endif
