/*
 * cutensorMp.h — multi-process distributed tensor contraction (cuTENSORMp C ABI), implemented for
 * AMD Instinct MI355X: one process per GPU, data exchanged with RCCL point-to-point transfers over xGMI.
 *
 * Reconstructed from the call sites of cutensorMp/cutensorMp_contraction.cu in NVIDIA/CUDALibrarySamples
 * (the header itself is not in the reference tree); every declaration cites the line that pins it.
 *
 * Distribution model (cutensorMp_contraction.cu:128-153, :471-483): mode i of extent E_i is cut into
 * nranksPerMode[i] contiguous blocks of ceil(E_i / nranksPerMode[i]) elements; grid cell (c_0, c_1, ...) —
 * linearised first mode fastest — lives on rank ranks[cell] (NULL: rank == cell index).  Every rank passes the
 * pointer of its own cell; the local buffer is a packed generalized-column-major tensor of the block extents
 * (the sample allocates exactly that, :438-446).  A tensor whose grid has a single cell is replicated: every
 * rank holds all of it ("the second input tensor ... is not distributed", :396).
 */
#ifndef CUTENSORMP_H_
#define CUTENSORMP_H_

#include <stdint.h>

#include <cutensor.h>   /* the sample uses cutensorStatus_t, cutensorDataType_t, CUTENSOR_OP_IDENTITY,
                           CUTENSOR_COMPUTE_DESC_32F and cutensorGetErrorString with only this header (:18, :453-454) */

#ifdef __cplusplus
extern "C" {
#endif

/* The communicator is RCCL's (`backend "nccl"` on ROCm); same declaration as <rccl/rccl.h>, repeated so that this
 * header does not depend on the include order of the caller (cutensorMp_contraction.cu:18 vs :28). */
typedef struct ncclComm* ncclComm_t;

typedef struct cutensorMpHandle*              cutensorMpHandle_t;
typedef struct cutensorMpTensorDescriptor*    cutensorMpTensorDescriptor_t;
typedef struct cutensorMpOperationDescriptor* cutensorMpOperationDescriptor_t;
typedef struct cutensorMpPlanPreference*      cutensorMpPlanPreference_t;
typedef struct cutensorMpPlan*                cutensorMpPlan_t;

typedef enum {
    CUTENSORMP_ALGO_DEFAULT = -1   /* cutensorMp_contraction.cu:494 */
} cutensorMpAlgo_t;

typedef enum {
    CUTENSORMP_PLAN_REQUIRED_WORKSPACE_DEVICE = 0,   /* uint64_t, cutensorMp_contraction.cu:506 */
    CUTENSORMP_PLAN_REQUIRED_WORKSPACE_HOST   = 1    /* uint64_t, cutensorMp_contraction.cu:508 */
} cutensorMpPlanAttribute_t;

/* cutensorMp_contraction.cu:470-471, :590 — one process per GPU; every call of this library enqueues on `stream`. */
cutensorStatus_t cutensorMpCreate(cutensorMpHandle_t* handle, ncclComm_t comm, int localDevice, cudaStream_t stream);
cutensorStatus_t cutensorMpDestroy(cutensorMpHandle_t handle);

/* cutensorMp_contraction.cu:474-483 — elementStride NULL = packed over the block extents; blockSize NULL =
 * ceil(extent / nranksPerMode) (block-cyclic layouts with several blocks per rank are NOT_SUPPORTED);
 * blockStride is ignored (one block per rank); ranks NULL = identity. */
cutensorStatus_t cutensorMpCreateTensorDescriptor(const cutensorMpHandle_t handle, cutensorMpTensorDescriptor_t* desc,
                                                  uint32_t numModes, const int64_t extent[], const int64_t elementStride[],
                                                  const int64_t blockSize[], const int64_t blockStride[],
                                                  const int64_t nranksPerMode[], uint32_t nranks, const int32_t ranks[],
                                                  cutensorDataType_t type);
cutensorStatus_t cutensorMpDestroyTensorDescriptor(cutensorMpTensorDescriptor_t desc);

/* cutensorMp_contraction.cu:485-488 — D = alpha opA(A) opB(B) + beta opC(C); D must be distributed like C. */
cutensorStatus_t cutensorMpCreateContraction(const cutensorMpHandle_t handle, cutensorMpOperationDescriptor_t* desc,
                                             const cutensorMpTensorDescriptor_t descA, const int32_t modesA[], cutensorOperator_t opA,
                                             const cutensorMpTensorDescriptor_t descB, const int32_t modesB[], cutensorOperator_t opB,
                                             const cutensorMpTensorDescriptor_t descC, const int32_t modesC[], cutensorOperator_t opC,
                                             const cutensorMpTensorDescriptor_t descD, const int32_t modesD[],
                                             const cutensorComputeDescriptor_t compute);
cutensorStatus_t cutensorMpDestroyOperationDescriptor(cutensorMpOperationDescriptor_t desc);

/* cutensorMp_contraction.cu:490-500 — budgets in bytes for the library's device / pinned-host scratch. */
cutensorStatus_t cutensorMpCreatePlanPreference(const cutensorMpHandle_t handle, cutensorMpPlanPreference_t* pref,
                                                cutensorMpAlgo_t algo, uint64_t workspaceSizeDeviceLimit,
                                                uint64_t workspaceSizeHostLimit);
cutensorStatus_t cutensorMpDestroyPlanPreference(cutensorMpPlanPreference_t pref);

/* cutensorMp_contraction.cu:502-509 — collective in the sense that every rank must build the same plan. */
cutensorStatus_t cutensorMpCreatePlan(const cutensorMpHandle_t handle, cutensorMpPlan_t* plan,
                                      const cutensorMpOperationDescriptor_t desc, const cutensorMpPlanPreference_t pref);
cutensorStatus_t cutensorMpPlanGetAttribute(const cutensorMpHandle_t handle, const cutensorMpPlan_t plan,
                                            cutensorMpPlanAttribute_t attr, void* buf, size_t sizeInBytes);
cutensorStatus_t cutensorMpDestroyPlan(cutensorMpPlan_t plan);

/* cutensorMp_contraction.cu:537-538 — A, B, C, D are this rank's local blocks; alpha / beta are host scalars of
 * the operation's scalar type; asynchronous on the handle's stream (the sample synchronises it, :119, :541). */
cutensorStatus_t cutensorMpContract(const cutensorMpHandle_t handle, const cutensorMpPlan_t plan, const void* alpha,
                                    const void* A, const void* B, const void* beta, const void* C, void* D,
                                    void* workspaceDevice, void* workspaceHost);

/* ---- engine-specific additions (not in the reference) ---------------------------------------------------------
 * A "local world" runs several ranks as threads of ONE process sharing one GPU, exchanging through device-to-device
 * copies instead of RCCL (RCCL refuses two ranks on one device).  It exists so that the multi-rank planning and
 * exchange logic can be exercised on a single-GPU machine; every rank's thread must be inside cutensorMpContract
 * at the same time, exactly as every process must be with RCCL. */
cutensorStatus_t ctamdMpLocalWorldCreate(void** world, int nranks);
cutensorStatus_t ctamdMpLocalWorldDestroy(void* world);
cutensorStatus_t ctamdMpCreateOnLocalWorld(cutensorMpHandle_t* handle, void* world, int rank, int localDevice,
                                           cudaStream_t stream);
/* JSON description of a plan (transfers, staging, local contraction) for tests and tools; returns bytes needed. */
size_t ctamdMpDescribePlan(const cutensorMpPlan_t plan, char* buf, size_t bufSize);

#ifdef __cplusplus
}
#endif

#endif /* CUTENSORMP_H_ */
