// explicit instantiations: K-slice GEMM for batched decode (gemm_kslice_kernel.hpp), W4, bf16, GPT=1
#include <algorithm>

#include "gemm_kslice_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_KSLICE_LAUNCH_SET(4, DIHIP_BF16, 1)
}  // namespace dihip
