/* TEST-HARNESS FIXTURE — einsum.cc:25 includes <cuda_fp16.hpp>; same names as cuda_fp16.h. */
#include "cuda_fp16.h"
