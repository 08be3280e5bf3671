/*
 * cutensorMg.h — single-process multi-GPU tensor contraction (cuTENSORMg C ABI), implemented for a
 * node of AMD Instinct MI355X GPUs connected by xGMI.
 *
 * Reconstructed from the call sites of cuTENSORMg/contraction_multi_gpu.cu in
 * NVIDIA/CUDALibrarySamples (the header itself is not in the reference tree); every declaration
 * cites the line that pins it.  One host thread drives all devices; tensors are block-cyclic
 * distributed: mode i is cut into blocks of blockSize[i] elements, block b of mode i belongs to
 * device-grid coordinate b % deviceCount[i], and grid cell (c_0, c_1, ...) — linearised first mode
 * fastest — is stored in the buffer devices[cell]/pointer[cell] (contraction_multi_gpu.cu:154-217,
 * :256-284).
 */
#ifndef CUTENSORMG_H_
#define CUTENSORMG_H_

#include <stdint.h>

#include <cutensor.h>   /* the Mg sample only includes this header and calls cutensorGetErrorString (contraction_multi_gpu.cu:43) */

#ifdef __cplusplus
extern "C" {
#endif

#define CUTENSOR_MG_DEVICE_HOST (-1)

typedef struct cutensorMgHandle*                cutensorMgHandle_t;
typedef struct cutensorMgTensorDescriptor*      cutensorMgTensorDescriptor_t;
typedef struct cutensorMgContractionDescriptor* cutensorMgContractionDescriptor_t;
typedef struct cutensorMgContractionFind*       cutensorMgContractionFind_t;
typedef struct cutensorMgContractionPlan*       cutensorMgContractionPlan_t;

typedef enum {
    CUTENSORMG_ALGO_DEFAULT = -1   /* contraction_multi_gpu.cu:237 */
} cutensorMgAlgo_t;

/* contraction_multi_gpu.cu:151, :383 */
cutensorStatus_t cutensorMgCreate(cutensorMgHandle_t* handle, uint32_t numDevices, const int32_t devices[]);
cutensorStatus_t cutensorMgDestroy(cutensorMgHandle_t handle);

/* contraction_multi_gpu.cu:195-197 — elementStride / blockStride NULL = packed */
cutensorStatus_t cutensorMgCreateTensorDescriptor(const cutensorMgHandle_t handle, cutensorMgTensorDescriptor_t* desc,
                                                  uint32_t numModes, const int64_t extent[],
                                                  const int64_t elementStride[], const int64_t blockSize[],
                                                  const int64_t blockStride[], const int32_t deviceCount[],
                                                  uint32_t numDevices, const int32_t devices[], cudaDataType_t type);
cutensorStatus_t cutensorMgDestroyTensorDescriptor(cutensorMgTensorDescriptor_t desc);

/* contraction_multi_gpu.cu:228-233 */
cutensorStatus_t cutensorMgCreateContractionDescriptor(const cutensorMgHandle_t handle, cutensorMgContractionDescriptor_t* desc,
                                                       const cutensorMgTensorDescriptor_t descA, const int32_t modesA[],
                                                       const cutensorMgTensorDescriptor_t descB, const int32_t modesB[],
                                                       const cutensorMgTensorDescriptor_t descC, const int32_t modesC[],
                                                       const cutensorMgTensorDescriptor_t descD, const int32_t modesD[],
                                                       cutensorComputeType_t compute);
cutensorStatus_t cutensorMgDestroyContractionDescriptor(cutensorMgContractionDescriptor_t desc);

/* contraction_multi_gpu.cu:236-237 */
cutensorStatus_t cutensorMgCreateContractionFind(const cutensorMgHandle_t handle, cutensorMgContractionFind_t* find,
                                                 const cutensorMgAlgo_t algo);
cutensorStatus_t cutensorMgDestroyContractionFind(cutensorMgContractionFind_t find);

/* contraction_multi_gpu.cu:241-242 — one size per handle device, plus a pinned-host size */
cutensorStatus_t cutensorMgContractionGetWorkspace(const cutensorMgHandle_t handle, const cutensorMgContractionDescriptor_t desc,
                                                   const cutensorMgContractionFind_t find, cutensorWorksizePreference_t preference,
                                                   int64_t deviceWorkspaceSize[], int64_t* hostWorkspaceSize);

/* contraction_multi_gpu.cu:249-250 */
cutensorStatus_t cutensorMgCreateContractionPlan(const cutensorMgHandle_t handle, cutensorMgContractionPlan_t* plan,
                                                 const cutensorMgContractionDescriptor_t desc, const cutensorMgContractionFind_t find,
                                                 const int64_t deviceWorkspaceSize[], int64_t hostWorkspaceSize);
cutensorStatus_t cutensorMgDestroyContractionPlan(cutensorMgContractionPlan_t plan);

/* contraction_multi_gpu.cu:328-332 — pointer arrays have one entry per grid cell of the respective
 * tensor; workspaceDevice / streams have one entry per handle device.  Asynchronous: returns after
 * enqueueing on the given streams; the caller synchronises every device (:334-338). */
cutensorStatus_t cutensorMgContraction(const cutensorMgHandle_t handle, const cutensorMgContractionPlan_t plan,
                                       const void* alpha, const void* A[], const void* B[], const void* beta,
                                       const void* C[], void* D[], void* workspaceDevice[], void* workspaceHost,
                                       cudaStream_t streams[]);

#ifdef __cplusplus
}
#endif

#endif /* CUTENSORMG_H_ */
