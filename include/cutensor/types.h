/*
 * cutensor/types.h — types of the cuTENSOR 2.x C ABI as exported by the MI355X-native engine.
 *
 * The reference repository (NVIDIA/CUDALibrarySamples) does not ship this header; every
 * declaration below is reconstructed from the call sites of the in-scope samples, cited as
 * <file>:<line> relative to the reference tree:
 *
 *   cuTENSOR/contraction.cu:37-40      cudaDataType_t / cutensorComputeDescriptor_t usage
 *   cuTENSOR/contraction.cu:123-223    handle / descriptor / plan object lifecycle
 *   cuTENSOR/reduction.cu:134          cutensorOperator_t (CUTENSOR_OP_ADD)
 *   cuTENSOR/python/einsum.h:39-67     cutensorDataType_t / CUTENSOR_R_* spelling
 *   cuTENSOR/python/einsum.h:31        CUTENSOR_STATUS_NOT_SUPPORTED special-casing
 *   cuTENSOR/contraction_plan_cache.cu:136  CUTENSOR_STATUS_IO_ERROR
 *   cuTENSORMg/contraction_multi_gpu.cu:223 cutensorComputeType_t (legacy 1.x enum used by Mg)
 *
 * Enumerator values are those of the public cuTENSOR 2.x headers so that objects built against
 * the original header stay binary compatible.
 *
 * The device runtime underneath is HIP.  The two runtime type names that appear in cuTENSOR
 * signatures (cudaStream_t, cudaDataType_t) are provided here as aliases of the HIP types, because
 * the ABI spells them that way; nothing else from that runtime is declared or used.
 */
#ifndef CUTENSOR_TYPES_H_
#define CUTENSOR_TYPES_H_

#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime_api.h>
#include <hip/library_types.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime type names mandated by the ABI spelling (contraction.cu:37, :245) ------------- */
typedef hipStream_t cudaStream_t;
typedef hipDataType cudaDataType_t;
#ifndef CUDA_R_32F
#define CUDA_R_16F  HIP_R_16F
#define CUDA_R_16BF HIP_R_16BF
#define CUDA_R_32F  HIP_R_32F
#define CUDA_R_64F  HIP_R_64F
#define CUDA_C_16F  HIP_C_16F
#define CUDA_C_32F  HIP_C_32F
#define CUDA_C_64F  HIP_C_64F
#define CUDA_R_8I   HIP_R_8I
#define CUDA_R_8U   HIP_R_8U
#define CUDA_R_32I  HIP_R_32I
#define CUDA_R_32U  HIP_R_32U
#endif

/* The samples pass cudaDataType_t (contraction.cu:136) and the binding passes
 * cutensorDataType_t (python/einsum.h:281-291) to the same entry point: one type, two spellings. */
typedef cudaDataType_t cutensorDataType_t;
#define CUTENSOR_R_16F  HIP_R_16F
#define CUTENSOR_C_16F  HIP_C_16F
#define CUTENSOR_R_16BF HIP_R_16BF
#define CUTENSOR_C_16BF HIP_C_16BF
#define CUTENSOR_R_32F  HIP_R_32F
#define CUTENSOR_C_32F  HIP_C_32F
#define CUTENSOR_R_64F  HIP_R_64F
#define CUTENSOR_C_64F  HIP_C_64F
#define CUTENSOR_R_8I   HIP_R_8I
#define CUTENSOR_R_8U   HIP_R_8U
#define CUTENSOR_R_32I  HIP_R_32I
#define CUTENSOR_R_32U  HIP_R_32U

/* ---- status codes (utils.cuh:35-39, python/einsum.h:31, contraction_plan_cache.cu:136) ------ */
typedef enum {
    CUTENSOR_STATUS_SUCCESS                = 0,
    CUTENSOR_STATUS_NOT_INITIALIZED        = 1,
    CUTENSOR_STATUS_ALLOC_FAILED           = 3,
    CUTENSOR_STATUS_INVALID_VALUE          = 7,
    CUTENSOR_STATUS_ARCH_MISMATCH          = 8,
    CUTENSOR_STATUS_MAPPING_ERROR          = 11,
    CUTENSOR_STATUS_EXECUTION_FAILED       = 13,
    CUTENSOR_STATUS_INTERNAL_ERROR         = 14,
    CUTENSOR_STATUS_NOT_SUPPORTED          = 15,
    CUTENSOR_STATUS_LICENSE_ERROR          = 16,
    CUTENSOR_STATUS_CUBLAS_ERROR           = 17,
    CUTENSOR_STATUS_CUDA_ERROR             = 18, /* here: an error reported by the HIP runtime */
    CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE = 19,
    CUTENSOR_STATUS_INSUFFICIENT_DRIVER    = 20,
    CUTENSOR_STATUS_IO_ERROR               = 21
} cutensorStatus_t;

/* ---- element-wise / reduction operators (contraction.cu:164, reduction.cu:134) -------------- */
typedef enum {
    CUTENSOR_OP_IDENTITY = 1,
    CUTENSOR_OP_SQRT     = 2,
    CUTENSOR_OP_ADD      = 3,
    CUTENSOR_OP_MUL      = 5,
    CUTENSOR_OP_MAX      = 6,
    CUTENSOR_OP_MIN      = 7,
    CUTENSOR_OP_RELU     = 8,
    CUTENSOR_OP_CONJ     = 9,
    CUTENSOR_OP_RCP      = 10,
    CUTENSOR_OP_SIGMOID  = 11,
    CUTENSOR_OP_TANH     = 12,
    CUTENSOR_OP_EXP      = 22,
    CUTENSOR_OP_LOG      = 23,
    CUTENSOR_OP_ABS      = 24,
    CUTENSOR_OP_NEG      = 25,
    CUTENSOR_OP_UNKNOWN  = 126
} cutensorOperator_t;

/* ---- algorithm selection (contraction.cu:192; contraction_jit.cu:212 uses ALGO_GETT) -------- */
typedef enum {
    CUTENSOR_ALGO_DEFAULT_PATIENT = -6, /* time every candidate kernel at plan creation */
    CUTENSOR_ALGO_GETT            = -4,
    CUTENSOR_ALGO_TGETT           = -3,
    CUTENSOR_ALGO_TTGT            = -2,
    CUTENSOR_ALGO_DEFAULT         = -1  /* heuristic; values >= 0 select candidate #algo directly */
} cutensorAlgo_t;

typedef enum {
    CUTENSOR_WORKSPACE_MIN     = 1,
    CUTENSOR_WORKSPACE_DEFAULT = 2,
    CUTENSOR_WORKSPACE_MAX     = 3
} cutensorWorksizePreference_t;

typedef enum {
    CUTENSOR_JIT_MODE_NONE    = 0,
    CUTENSOR_JIT_MODE_DEFAULT = 1
} cutensorJitMode_t;

typedef enum {
    CUTENSOR_AUTOTUNE_MODE_NONE        = 0,
    CUTENSOR_AUTOTUNE_MODE_INCREMENTAL = 1
} cutensorAutotuneMode_t;

typedef enum {
    CUTENSOR_CACHE_MODE_NONE     = 0,
    CUTENSOR_CACHE_MODE_PEDANTIC = 1
} cutensorCacheMode_t;

/* contraction.cu:176-180 queries SCALAR_TYPE; contraction_jit.cu:379-383 queries FLOPS */
typedef enum {
    CUTENSOR_OPERATION_DESCRIPTOR_TAG           = 0, /* int32_t */
    CUTENSOR_OPERATION_DESCRIPTOR_SCALAR_TYPE   = 1, /* cutensorDataType_t */
    CUTENSOR_OPERATION_DESCRIPTOR_FLOPS         = 2, /* float */
    CUTENSOR_OPERATION_DESCRIPTOR_MOVED_BYTES   = 3, /* float */
    CUTENSOR_OPERATION_DESCRIPTOR_PADDING_LEFT  = 4,
    CUTENSOR_OPERATION_DESCRIPTOR_PADDING_RIGHT = 5,
    CUTENSOR_OPERATION_DESCRIPTOR_PADDING_VALUE = 6
} cutensorOperationDescriptorAttribute_t;

typedef enum {
    CUTENSOR_PLAN_PREFERENCE_AUTOTUNE_MODE     = 0, /* cutensorAutotuneMode_t */
    CUTENSOR_PLAN_PREFERENCE_CACHE_MODE        = 1, /* cutensorCacheMode_t */
    CUTENSOR_PLAN_PREFERENCE_INCREMENTAL_COUNT = 2, /* int32_t */
    CUTENSOR_PLAN_PREFERENCE_ALGO              = 3, /* cutensorAlgo_t */
    CUTENSOR_PLAN_PREFERENCE_KERNEL_RANK       = 4, /* int32_t */
    CUTENSOR_PLAN_PREFERENCE_JIT               = 5, /* cutensorJitMode_t */
    /* Engine extension (not a cuTENSOR attribute; values of 1000 and above are this library's own): int32_t, non-zero = "the operands of
     * this contraction are streamed" — every call reads operands that are not resident in the 256-MiB Infinity Cache (a different (A, B)
     * each call, or tensors far larger than the cache).  The planner then ranks the nontemporal-load twins of the streaming kernels for
     * read-once problems of ANY size (the headline einsum, 201 MB of operands: +3.6 % from HBM, -5.5 % when the same operands are
     * re-contracted out of the cache — a library cannot know which it will be, a caller can; profiles/r03_headline_nt.txt). */
    CUTENSOR_AMD_PLAN_PREFERENCE_OPERANDS_STREAMED = 1000
} cutensorPlanPreferenceAttribute_t;

/* contraction.cu:231-235 */
typedef enum {
    CUTENSOR_PLAN_REQUIRED_WORKSPACE = 0 /* uint64_t */
} cutensorPlanAttribute_t;

/* Legacy 1.x compute-type enum; only the Mg entry points still take it
 * (contraction_multi_gpu.cu:223 passes CUTENSOR_COMPUTE_32F). */
typedef enum {
    CUTENSOR_COMPUTE_16F    = (1U << 0U),
    CUTENSOR_COMPUTE_16BF   = (1U << 10U),
    CUTENSOR_COMPUTE_TF32   = (1U << 12U),
    CUTENSOR_COMPUTE_3XTF32 = (1U << 13U),
    CUTENSOR_COMPUTE_32F    = (1U << 2U),
    CUTENSOR_COMPUTE_64F    = (1U << 4U)
} cutensorComputeType_t;

/* ---- opaque objects (all pointer-sized; created by cutensorCreate*, freed by cutensorDestroy*) */
typedef struct cutensorHandle*              cutensorHandle_t;
typedef struct cutensorTensorDescriptor*    cutensorTensorDescriptor_t;
typedef struct cutensorOperationDescriptor* cutensorOperationDescriptor_t;
typedef struct cutensorPlanPreference*      cutensorPlanPreference_t;
typedef struct cutensorPlan*                cutensorPlan_t;
typedef struct cutensorBlockSparseTensorDescriptor* cutensorBlockSparseTensorDescriptor_t;   /* blocksparse.cu:64 */

/* Compute descriptors are opaque pointers to library-owned constants and are used as values
 * (contraction.cu:40, einsum.cu:39,46,53). */
typedef const struct cutensorComputeDescriptor* cutensorComputeDescriptor_t;

#ifdef __cplusplus
}
#endif

#endif /* CUTENSOR_TYPES_H_ */
