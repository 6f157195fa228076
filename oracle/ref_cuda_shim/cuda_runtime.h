// oracle/ref_cuda_shim: the reference's span_attn.h includes <cuda_runtime.h> for cudaStream_t / cudaDeviceProp in its host API;
// the codec headers compiled here use none of it.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <hip/hip_runtime.h>
typedef hipStream_t cudaStream_t;
typedef hipDeviceProp_t cudaDeviceProp;
