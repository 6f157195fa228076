// oracle/codec_ref.hip -- TEST INFRASTRUCTURE ONLY (never linked into, loaded by or shipped with the product).
//
// Runs the REFERENCE'S OWN quantised-KV codec -- span::qcache::QuantParam<I8 | U4, T>::Builder (per-head min / max -> scale, zero)
// and ::Quant, included from /root/reference/span-attention/src/cache_quant/impl_{i8,u4}.cuh as they lie -- over rows of 128
// values, the way the reference's append kernel calls it: one 32-lane warp per (token, head), 4 elements per lane
// (csrc/core/kernel/cuda/cache/decoder_cache_append.cuh:33-86 transposeQuantAppend: `builder(regs)`, `param.Quant(q[i], regs + i * U)`,
// lane 0 writes {zero, scale}).  The outputs are compared with the product's span bytes in tests/test_gpu_kv_codec_ref.py.
//
// __fdividef: utils.cuh:28-32 divides floats with CUDA's approximate __fdividef, which HIP does not have and whose bits on
// NVIDIA hardware cannot be reproduced here.  The library is built twice (oracle/Makefile): REF_FDIV_RCP undefined -> hipcc's
// own __fdividef = IEEE division (what the product and oracle/kv_codec.py do); defined -> a * v_rcp_f32(b) (1 ulp reciprocal + rounding of the
// product: an approximate division of the same error class, <= 2 ulp).  The test reports how many codes / parameters move
// between the two: that set bounds what any <= 2-ulp division can change.
#include <hip/hip_runtime.h>
#include <cstdint>

// (hipcc's own __fdividef, __clang_hip_math.h:234, IS `x / y`: the first build is the reference's code as hipcc compiles it)
#ifdef REF_FDIV_RCP
__device__ __forceinline__ float ref_fdividef_rcp(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#define __fdividef ref_fdividef_rcp
#endif

#include "impl_i8.cuh"   // -I/root/reference/span-attention/src/cache_quant (oracle/Makefile)
#include "impl_u4.cuh"

namespace {

template <span::QuantMode MODE>
__global__ __launch_bounds__(32) void ref_codec_rows(const float* x, unsigned char* q, float* params) {
  using Param = span::qcache::QuantParam<MODE, float>;
  constexpr int U = span::qcache::QCacheConfig<MODE, float>::UNDERLYING_SIZE;
  const int row = blockIdx.x;
  span::PackT<4, float> pack;
#pragma unroll
  for (int i = 0; i < 4; ++i) pack.data[i] = x[(size_t)row * 128 + threadIdx.x * 4 + i];
  const float(&regs)[4] = pack.data;
  const typename Param::template Builder<32, 4, 128> builder;
  const Param param = builder(regs);
#pragma unroll
  for (int i = 0; i < 4 / U; ++i) {
    typename span::qcache::QCacheConfig<MODE, float>::QuantT qv;
    param.Quant(qv, regs + i * U);
    if constexpr (MODE == span::QuantMode::U4) q[(size_t)row * 64 + threadIdx.x * 2 + i] = qv.raw;
    else q[(size_t)row * 128 + threadIdx.x * 4 + i] = (unsigned char)qv;
  }
  if (threadIdx.x == 0) {
    params[(size_t)row * 2] = param.zero;
    params[(size_t)row * 2 + 1] = param.scale;
  }
}

}  // namespace

// x: device f32 [rows, 128]; q: device bytes [rows, 64 (u4) | 128 (i8)]; params: device f32 [rows, 2] = {zero, scale}.  mode: 1 = I8, 2 = U4.
extern "C" int ref_codec_quantize(int mode, const float* x, unsigned char* q, float* params, int rows, void* stream) {
  if (rows <= 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (mode == 1) hipLaunchKernelGGL(ref_codec_rows<span::QuantMode::I8>, dim3(rows), dim3(32), 0, s, x, q, params);
  else if (mode == 2) hipLaunchKernelGGL(ref_codec_rows<span::QuantMode::U4>, dim3(rows), dim3(32), 0, s, x, q, params);
  else return 2;
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
extern "C" const char* ref_codec_division(void) {
#ifdef REF_FDIV_RCP
  return "a * v_rcp_f32(b)";
#else
  return "ieee";
#endif
}
