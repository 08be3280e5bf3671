/* TEST-HARNESS FIXTURE (see cuda_runtime.h in this directory): cutensorMp/cutensorMp_contraction.cu:28 includes
 * <nccl.h>; on ROCm the same API is RCCL's. */
#ifndef SAMPLE_COMPAT_NCCL_H_
#define SAMPLE_COMPAT_NCCL_H_
#include <rccl/rccl.h>
#endif
