// oracle/ref_cuda_shim stand-in for span-attention/src/utils/pack.cuh (PTX bfe / mov byte extraction): the two templates the
// codec headers use, with plain element access.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstdint>
#include "common/func_modifier.h"
namespace span {
template <int PACK_SIZE, typename T>
struct alignas(PACK_SIZE * sizeof(T)) WordPackT {
  T data[PACK_SIZE];
  template <typename ComputeT>
  DEVICE_FUNC void Unpack(ComputeT (&ret)[PACK_SIZE]) const {
#pragma unroll
    for (int i = 0; i < PACK_SIZE; ++i) ret[i] = static_cast<ComputeT>(data[i]);
  }
};
template <int PACK_SIZE, typename T>
struct alignas(PACK_SIZE * sizeof(T)) PackT {
  T data[PACK_SIZE];
};
}  // namespace span
