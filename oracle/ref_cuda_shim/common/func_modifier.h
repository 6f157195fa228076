// oracle/ref_cuda_shim stand-in for span-attention/src/common/func_modifier.h (which keys on __CUDACC__).  TEST INFRASTRUCTURE ONLY.
#pragma once
#define DEVICE_FUNC __device__ __forceinline__
#define HOST_DEVICE_FUNC __host__ __device__ inline
