/*
 * TEST-HARNESS FIXTURE — not part of the engine, not shipped in the library.
 *
 * The reference samples (cuTENSOR/contraction.cu, einsum.cu, reduction.cu, elementwise_permute.cu,
 * cuTENSORMg/contraction_multi_gpu.cu) include <cuda_runtime.h> and call ~30 runtime entry points
 * directly (cuTENSOR/utils.cuh:51-71, :167-191).  To prove that the UNMODIFIED sample sources
 * compile and link against libcutensor.so / libcutensorMg.so, oracle/build_ref_samples.sh puts this
 * directory on the include path; the names below are spelled onto the HIP runtime the samples then
 * run on.  The engine itself (cudalibrarysamples_amd/csrc) never includes this file.
 */
#ifndef SAMPLE_COMPAT_CUDA_RUNTIME_H_
#define SAMPLE_COMPAT_CUDA_RUNTIME_H_

#include <hip/hip_runtime.h>
#include <hip/library_types.h>

typedef hipError_t  cudaError_t;
typedef hipStream_t cudaStream_t;
typedef hipEvent_t  cudaEvent_t;
typedef hipDataType cudaDataType_t;
struct cudaDeviceProp : public hipDeviceProp_t {};   /* the Mg sample writes `struct cudaDeviceProp` (contraction_multi_gpu.cu:108) */

#define cudaSuccess                 hipSuccess
#define cudaMalloc                  hipMalloc
#define cudaFree                    hipFree
#define cudaMallocHost              hipHostMalloc
#define cudaFreeHost                hipHostFree
#define cudaMemcpy                  hipMemcpy
#define cudaMemset                  hipMemset
#define cudaMemGetInfo              hipMemGetInfo            /* torch/einsum.cc:70 */
#define cudaMemcpyAsync             hipMemcpyAsync
#define cudaMemcpy2DAsync           hipMemcpy2DAsync
#define cudaMemcpyHostToDevice      hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost      hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice    hipMemcpyDeviceToDevice
#define cudaMemcpyDefault           hipMemcpyDefault
#define cudaStreamCreate            hipStreamCreate
#define cudaStreamDestroy           hipStreamDestroy
#define cudaStreamSynchronize       hipStreamSynchronize
#define cudaEventCreate             hipEventCreate
#define cudaEventDestroy            hipEventDestroy
#define cudaEventRecord             hipEventRecord
#define cudaEventSynchronize        hipEventSynchronize
#define cudaEventElapsedTime        hipEventElapsedTime
#define cudaDeviceSynchronize       hipDeviceSynchronize
#define cudaGetErrorString          hipGetErrorString
#define cudaGetErrorName            hipGetErrorName
#define cudaSetDevice               hipSetDevice
#define cudaGetDevice               hipGetDevice
#define cudaGetDeviceCount          hipGetDeviceCount
#define cudaGetDeviceProperties     hipGetDeviceProperties
#define cudaDeviceGetAttribute      hipDeviceGetAttribute
#define cudaDevAttrClockRate        hipDeviceAttributeClockRate
#define cudaDevAttrMemoryClockRate  hipDeviceAttributeMemoryClockRate

#endif
