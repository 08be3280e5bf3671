/*
 * cutensor.h — the cuTENSOR 2.x C ABI, implemented natively for AMD Instinct MI355X (gfx950).
 *
 * This is the drop-in boundary of the engine: plain C entry points, opaque pointer-sized handles,
 * host pointers for scalars, device pointers for tensors, a HIP stream for ordering.  No C++ or
 * framework types cross it.  Each declaration cites the call site in NVIDIA/CUDALibrarySamples
 * that pins its shape (paths relative to the reference tree); the reference does not contain the
 * header itself.
 *
 * Conventions (all evidenced by the samples):
 *   - every Create* returns an object the caller destroys; descriptors may be destroyed as soon as
 *     the plan is built (python/einsum.h:394-397) — plans copy what they need;
 *   - Destroy*(NULL) is tolerated (python/einsum.h:302,396);
 *   - extents/strides are int64_t, modes are int32_t labels, stride == NULL means packed
 *     generalized column-major, first listed mode fastest (blocksparse.cu:80-81);
 *   - alpha/beta are HOST pointers of the scalar type reported by
 *     CUTENSOR_OPERATION_DESCRIPTOR_SCALAR_TYPE (fp32 for fp32/fp16/bf16 data: einsum.cu:47,54);
 *   - the library never allocates user-visible device memory; workspace is caller-provided and
 *     may be NULL/0 (contraction_plan_cache.cu:252);
 *   - execution is asynchronous on the caller's stream, the NULL stream included
 *     (elementwise_permute.cu:200);
 *   - errors are returned as cutensorStatus_t; nothing throws across this boundary.
 */
#ifndef CUTENSOR_H_
#define CUTENSOR_H_

#include <stdint.h>
#include <stdio.h>

#include <cutensor/types.h>

#define CUTENSOR_MAJOR 2
#define CUTENSOR_MINOR 2
#define CUTENSOR_PATCH 0
#define CUTENSOR_VERSION (CUTENSOR_MAJOR * 10000 + CUTENSOR_MINOR * 100 + CUTENSOR_PATCH)
/* This header belongs to the MI355X engine: lets a caller guard the engine-prefixed extras (CUTENSOR_AMD_PLAN_PREFERENCE_*,
 * ctamd* entry points) that cuTENSOR proper does not have. */
#define CUTENSOR_AMD 1

#ifdef __cplusplus
extern "C" {
#endif

/* ---- compute descriptors: exported data symbols (einsum.cu:39,46,53; contraction.cu:40) ----- */
extern const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_16F;
extern const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_16BF;
extern const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_TF32;
extern const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_3XTF32;
extern const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_32F;
extern const cutensorComputeDescriptor_t CUTENSOR_COMPUTE_DESC_64F;

/* ---- library handle (contraction.cu:123-124) ------------------------------------------------ */
cutensorStatus_t cutensorCreate(cutensorHandle_t* handle);
cutensorStatus_t cutensorDestroy(cutensorHandle_t handle);

/* einsum.cu:445 — number of plans memoised per handle (0 disables the cache) */
cutensorStatus_t cutensorHandleResizePlanCache(cutensorHandle_t handle, const uint32_t numEntries);
/* contraction_plan_cache.cu:132-148, 324-337 */
cutensorStatus_t cutensorHandleWritePlanCacheToFile(const cutensorHandle_t handle, const char filename[]);
cutensorStatus_t cutensorHandleReadPlanCacheFromFile(cutensorHandle_t handle, const char filename[],
                                                     uint32_t* numCachelinesRead);

/* ---- tensor descriptor (contraction.cu:131-137) --------------------------------------------- */
cutensorStatus_t cutensorCreateTensorDescriptor(const cutensorHandle_t      handle,
                                                cutensorTensorDescriptor_t* desc,
                                                const uint32_t              numModes,
                                                const int64_t               extent[],
                                                const int64_t               stride[], /* NULL = packed */
                                                cutensorDataType_t          dataType,
                                                uint32_t                    alignmentRequirement);
cutensorStatus_t cutensorDestroyTensorDescriptor(cutensorTensorDescriptor_t desc);

/* ---- operation descriptors ------------------------------------------------------------------ */
/* D = alpha * opA(A) * opB(B) + beta * opC(C)                          contraction.cu:162-168 */
cutensorStatus_t cutensorCreateContraction(const cutensorHandle_t           handle,
                                           cutensorOperationDescriptor_t*   desc,
                                           const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                           const cutensorTensorDescriptor_t descB, const int32_t modeB[], cutensorOperator_t opB,
                                           const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                           const cutensorTensorDescriptor_t descD, const int32_t modeD[],
                                           const cutensorComputeDescriptor_t descCompute);

/* D = alpha * reduce_opReduce(opA(A)) + beta * opC(C)                      reduction.cu:141-146 */
cutensorStatus_t cutensorCreateReduction(const cutensorHandle_t           handle,
                                         cutensorOperationDescriptor_t*   desc,
                                         const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                         const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                         const cutensorTensorDescriptor_t descD, const int32_t modeD[],
                                         cutensorOperator_t opReduce,
                                         const cutensorComputeDescriptor_t descCompute);

/* B = alpha * opA(perm(A))                                       elementwise_permute.cu:142-149 */
cutensorStatus_t cutensorCreatePermutation(const cutensorHandle_t           handle,
                                           cutensorOperationDescriptor_t*   desc,
                                           const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                           const cutensorTensorDescriptor_t descB, const int32_t modeB[],
                                           const cutensorComputeDescriptor_t descCompute);

/* D = opAC(alpha * opA(perm(A)), gamma * opC(perm(C)))            elementwise_binary.cu:149-153 */
cutensorStatus_t cutensorCreateElementwiseBinary(const cutensorHandle_t           handle,
                                                 cutensorOperationDescriptor_t*   desc,
                                                 const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                                 const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                                 const cutensorTensorDescriptor_t descD, const int32_t modeD[],
                                                 cutensorOperator_t opAC,
                                                 const cutensorComputeDescriptor_t descCompute);

cutensorStatus_t cutensorDestroyOperationDescriptor(cutensorOperationDescriptor_t desc);

/* contraction.cu:176-180 (SCALAR_TYPE), contraction_jit.cu:379-383 (FLOPS) */
cutensorStatus_t cutensorOperationDescriptorGetAttribute(const cutensorHandle_t handle,
                                                         cutensorOperationDescriptor_t desc,
                                                         cutensorOperationDescriptorAttribute_t attr,
                                                         void* buf, size_t sizeInBytes);
cutensorStatus_t cutensorOperationDescriptorSetAttribute(const cutensorHandle_t handle,
                                                         cutensorOperationDescriptor_t desc,
                                                         cutensorOperationDescriptorAttribute_t attr,
                                                         const void* buf, size_t sizeInBytes);

/* ---- plan preference / workspace / plan (contraction.cu:194-235) ----------------------------- */
cutensorStatus_t cutensorCreatePlanPreference(const cutensorHandle_t    handle,
                                              cutensorPlanPreference_t* pref,
                                              cutensorAlgo_t            algo,
                                              cutensorJitMode_t         jitMode);
cutensorStatus_t cutensorDestroyPlanPreference(cutensorPlanPreference_t pref);
/* contraction_plan_cache.cu:215-237 */
cutensorStatus_t cutensorPlanPreferenceSetAttribute(const cutensorHandle_t handle,
                                                    cutensorPlanPreference_t pref,
                                                    cutensorPlanPreferenceAttribute_t attr,
                                                    const void* buf, size_t sizeInBytes);

cutensorStatus_t cutensorEstimateWorkspaceSize(const cutensorHandle_t              handle,
                                               const cutensorOperationDescriptor_t desc,
                                               const cutensorPlanPreference_t      planPref,
                                               const cutensorWorksizePreference_t  workspacePref,
                                               uint64_t*                           workspaceSizeEstimate);

cutensorStatus_t cutensorCreatePlan(const cutensorHandle_t              handle,
                                    cutensorPlan_t*                     plan,
                                    const cutensorOperationDescriptor_t desc,
                                    const cutensorPlanPreference_t      pref, /* NULL = defaults */
                                    uint64_t                            workspaceSizeLimit);
cutensorStatus_t cutensorDestroyPlan(cutensorPlan_t plan);
cutensorStatus_t cutensorPlanGetAttribute(const cutensorHandle_t handle,
                                          const cutensorPlan_t   plan,
                                          cutensorPlanAttribute_t attr,
                                          void* buf, size_t sizeInBytes);

/* ---- execution: the hot path ----------------------------------------------------------------- */
/* contraction.cu:261-265, einsum.cu:334-338 */
cutensorStatus_t cutensorContract(const cutensorHandle_t handle, const cutensorPlan_t plan,
                                  const void* alpha, const void* A, const void* B,
                                  const void* beta,  const void* C, void* D,
                                  void* workspace, uint64_t workspaceSize, cudaStream_t stream);
/* reduction.cu:219-222, einsum.cu:369-372 */
cutensorStatus_t cutensorReduce(const cutensorHandle_t handle, const cutensorPlan_t plan,
                                const void* alpha, const void* A,
                                const void* beta,  const void* C, void* D,
                                void* workspace, uint64_t workspaceSize, cudaStream_t stream);
/* elementwise_permute.cu:198-200 */
cutensorStatus_t cutensorPermute(const cutensorHandle_t handle, const cutensorPlan_t plan,
                                 const void* alpha, const void* A, void* B, const cudaStream_t stream);
/* elementwise_binary.cu:202-205 */
cutensorStatus_t cutensorElementwiseBinaryExecute(const cutensorHandle_t handle, const cutensorPlan_t plan,
                                                  const void* alpha, const void* A,
                                                  const void* gamma, const void* C, void* D,
                                                  cudaStream_t stream);

/* elementwise_trinary.cu:174-182: D = opABC(opAB(alpha opA(A), beta opB(B)), gamma opC(C)) */
cutensorStatus_t cutensorCreateElementwiseTrinary(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                                  const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                                  const cutensorTensorDescriptor_t descB, const int32_t modeB[], cutensorOperator_t opB,
                                                  const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                                  const cutensorTensorDescriptor_t descD, const int32_t modeD[],
                                                  cutensorOperator_t opAB, cutensorOperator_t opABC,
                                                  const cutensorComputeDescriptor_t descCompute);
/* elementwise_trinary.cu:223-227 */
cutensorStatus_t cutensorElementwiseTrinaryExecute(const cutensorHandle_t handle, const cutensorPlan_t plan,
                                                   const void* alpha, const void* A, const void* beta, const void* B,
                                                   const void* gamma, const void* C, void* D, cudaStream_t stream);
/* contraction_trinary.cu:191-198: E = alpha * opA(A) * opB(B) * opC(C) + beta * opD(D) */
cutensorStatus_t cutensorCreateContractionTrinary(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                                  const cutensorTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                                  const cutensorTensorDescriptor_t descB, const int32_t modeB[], cutensorOperator_t opB,
                                                  const cutensorTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                                  const cutensorTensorDescriptor_t descD, const int32_t modeD[], cutensorOperator_t opD,
                                                  const cutensorTensorDescriptor_t descE, const int32_t modeE[],
                                                  const cutensorComputeDescriptor_t descCompute);
/* contraction_trinary.cu:290-294 */
cutensorStatus_t cutensorContractTrinary(const cutensorHandle_t handle, const cutensorPlan_t plan, const void* alpha,
                                         const void* A, const void* B, const void* C, const void* beta, const void* D, void* E,
                                         void* workspace, uint64_t workspaceSize, cudaStream_t stream);
/* ---- block-sparse tensors (blocksparse.cu) ----------------------------------------------------------
 * A tensor is a set of dense blocks: every mode is cut into sections (extent[] lists the section extents mode by
 * mode), a block is addressed by one section index per mode (nonZeroCoordinates[block * numModes + mode]) and lives
 * behind its own device pointer; stride == NULL: every block packed column-major. blocksparse.cu:102-107 */
cutensorStatus_t cutensorCreateBlockSparseTensorDescriptor(cutensorHandle_t handle, cutensorBlockSparseTensorDescriptor_t* desc,
                                                           const uint32_t numModes, const uint64_t numNonZeroBlocks,
                                                           const uint32_t numSectionsPerMode[], const int64_t extent[],
                                                           const int32_t nonZeroCoordinates[], const int64_t stride[],
                                                           cudaDataType_t dataType);
cutensorStatus_t cutensorDestroyBlockSparseTensorDescriptor(cutensorBlockSparseTensorDescriptor_t desc);
/* blocksparse.cu:177-182 */
cutensorStatus_t cutensorCreateBlockSparseContraction(const cutensorHandle_t handle, cutensorOperationDescriptor_t* desc,
                                                      const cutensorBlockSparseTensorDescriptor_t descA, const int32_t modeA[], cutensorOperator_t opA,
                                                      const cutensorBlockSparseTensorDescriptor_t descB, const int32_t modeB[], cutensorOperator_t opB,
                                                      const cutensorBlockSparseTensorDescriptor_t descC, const int32_t modeC[], cutensorOperator_t opC,
                                                      const cutensorBlockSparseTensorDescriptor_t descD, const int32_t modeD[],
                                                      const cutensorComputeDescriptor_t descCompute);
/* blocksparse.cu:206-209: A, B, C, D are host arrays of device pointers, one per non-zero block */
cutensorStatus_t cutensorBlockSparseContract(const cutensorHandle_t handle, const cutensorPlan_t plan, const void* alpha,
                                             const void* const A[], const void* const B[], const void* beta,
                                             const void* const C[], void* const D[], void* workspace, uint64_t workspaceSize,
                                             cudaStream_t stream);
/* contraction_jit.cu:134 / :398 — accepted for source compatibility; there is no run-time code generation */
cutensorStatus_t cutensorReadKernelCacheFromFile(cutensorHandle_t handle, const char filename[]);
cutensorStatus_t cutensorWriteKernelCacheToFile(const cutensorHandle_t handle, const char filename[]);

/* ---- misc (utils.cuh:38) --------------------------------------------------------------------- */
const char* cutensorGetErrorString(const cutensorStatus_t error);
size_t      cutensorGetVersion(void);

#ifdef __cplusplus
}
#endif

#endif /* CUTENSOR_H_ */
