// explicit instantiations: context-phase GEMM (gemm_prefill_kernel.hpp), W4, f16
#include "gemm_prefill_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_PREFILL_LAUNCH_SET(4, DIHIP_F16)
}  // namespace dihip
