// explicit instantiations: context-phase GEMM (gemm_prefill_kernel.hpp), W8, f16
#include "gemm_prefill_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_PREFILL_LAUNCH_SET(8, DIHIP_F16)
}  // namespace dihip
