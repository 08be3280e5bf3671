/*
 * TEST-HARNESS FIXTURE — not part of the engine, not shipped in the library.
 *
 * The reference's PyTorch binding (cuTENSOR/python/cutensor/torch/einsum.cc:19-22) includes
 * <ATen/cuda/CUDAContext.h> and calls at::cuda::getCurrentCUDAStream(), at::cuda::current_device() and
 * at::cuda::CUDACachingAllocator::getDeviceStats() (einsum.cc:59-60,126).  torch-ROCm ships those under the
 * ATen/hip, c10/hip spellings; oracle/build_ref_torch_binding.sh puts THIS directory ahead of torch's own
 * include path so the UNMODIFIED einsum.cc compiles against the ROCm build of torch.
 */
#pragma once
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
#include <c10/hip/HIPFunctions.h>

namespace at { namespace cuda {
inline c10::hip::HIPStreamMasqueradingAsCUDA getCurrentCUDAStream(c10::DeviceIndex device_index = -1) {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(device_index);
}
namespace CUDACachingAllocator = c10::hip::HIPCachingAllocatorMasqueradingAsCUDA;
}}  // namespace at::cuda
