// oracle/ref_cuda_shim stand-in for span-attention/src/utils/shuffle.cuh: the reference spells its warp shuffles as PTX
// (shfl.sync.bfly / shfl.sync.idx with a width); here the HIP shuffles with the same (laneMask | srcLane, width) meaning.  The
// reference's warps are 32 lanes: the shim kernel launches 32-thread blocks, width <= 32 keeps every exchange inside them.
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstdint>
#include "common/func_modifier.h"
namespace span {
template <typename T>
DEVICE_FUNC T ShflBfly(uint32_t, const T& var, uint32_t laneMask, uint32_t width) {
  return __shfl_xor(var, (int)laneMask, (int)width);
}
template <typename T>
DEVICE_FUNC T ShflIdx(uint32_t, const T& var, uint32_t srcLane, uint32_t width) {
  return __shfl(var, (int)srcLane, (int)width);
}
}  // namespace span
