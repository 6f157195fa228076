// explicit instantiations: small-batch decode GEMM (gemv_batch_kernel.hpp), W8, bf16, GPT=0
#include "gemv_batch_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_GEMB_LAUNCH_SET(8, DIHIP_BF16, 0)
}  // namespace dihip
