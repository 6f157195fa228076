// gemm_lowp_launch.hpp -- launch thunks of the weight-only GEMM kernels.  The template
// instantiations are spread over several translation units (gemm_lowp_inst_*.hip) so that the
// gfx950 compile parallelises.
#pragma once
#include "gemm_lowp_kernel.hpp"

namespace dihip {

template <int WBITS, int FT, int MT, int NT, int PRO, int EPI>
hipError_t launch_gemm_lowp(const GemmArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream);

#define DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, MT, NT, PRO, EPI)                                   \
  template <>                                                                                   \
  hipError_t launch_gemm_lowp<WBITS, FT, MT, NT, PRO, EPI>(const GemmArgs& a, dim3 grid,        \
                                                           size_t lds_bytes, hipStream_t s) {   \
    auto kern = gemm_lowp_kernel<WBITS, FT, MT, NT, PRO, EPI>;                                  \
    if (lds_bytes > 64 * 1024) {                                                                \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                         (int)lds_bytes);                                       \
      if (e != hipSuccess) return e;                                                            \
    }                                                                                           \
    hipLaunchKernelGGL(kern, grid, dim3(GEMM_THREADS), lds_bytes, s, a);                        \
    return hipGetLastError();                                                                   \
  }

// every (MT, NT) shape used by the dispatcher for one (WBITS, FT)
#define DIHIP_DEFINE_GEMM_LAUNCH_SET_STD(WBITS, FT)                  \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 1, 2, PRO_PLAIN, EPI_STD)      \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 1, 4, PRO_PLAIN, EPI_STD)      \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 2, 2, PRO_PLAIN, EPI_STD)      \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 2, 4, PRO_PLAIN, EPI_STD)

#define DIHIP_DEFINE_GEMM_LAUNCH_SET_FUSED(WBITS, FT)                \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 1, 2, PRO_RMSNORM, EPI_STD)    \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 1, 4, PRO_RMSNORM, EPI_STD)    \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 1, 4, PRO_RMSNORM, EPI_SWIGLU) \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 1, 4, PRO_PLAIN, EPI_SWIGLU)   \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 2, 4, PRO_PLAIN, EPI_SWIGLU)   \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 1, 2, PRO_PLAIN, EPI_ADDTO)    \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 1, 4, PRO_PLAIN, EPI_ADDTO)    \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 2, 2, PRO_PLAIN, EPI_ADDTO)    \
  DIHIP_DEFINE_GEMM_LAUNCH(WBITS, FT, 2, 4, PRO_PLAIN, EPI_ADDTO)

// unquantised (W16) weights: lm_head = rmsnorm prologue (M <= 4) or plain, f32 logits out
#define DIHIP_DEFINE_GEMM_LAUNCH_SET_DENSE(FT)                       \
  DIHIP_DEFINE_GEMM_LAUNCH(16, FT, 1, 4, PRO_RMSNORM, EPI_ADDTO)     \
  DIHIP_DEFINE_GEMM_LAUNCH(16, FT, 1, 4, PRO_PLAIN, EPI_ADDTO)       \
  DIHIP_DEFINE_GEMM_LAUNCH(16, FT, 2, 4, PRO_PLAIN, EPI_ADDTO)       \
  DIHIP_DEFINE_GEMM_LAUNCH(16, FT, 1, 4, PRO_PLAIN, EPI_STD)         \
  DIHIP_DEFINE_GEMM_LAUNCH(16, FT, 2, 4, PRO_PLAIN, EPI_STD)

}  // namespace dihip
