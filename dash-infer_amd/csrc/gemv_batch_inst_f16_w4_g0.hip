// explicit instantiations: small-batch decode GEMM (gemv_batch_kernel.hpp), W4, f16, GPT=0
#include "gemv_batch_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_GEMB_LAUNCH_SET(4, DIHIP_F16, 0)
}  // namespace dihip
