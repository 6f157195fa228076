// explicit instantiations: K-slice GEMM for batched decode, deferred-RMSNorm consumer (gemm_kslice_kernel.hpp, RS = 1), W4, bf16, GPT=0
#include <algorithm>

#include "gemm_kslice_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_KSLICE_RS_LAUNCH_SET(4, 0)
}  // namespace dihip
