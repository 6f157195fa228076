// explicit instantiations: K-slice GEMM for batched decode (gemm_kslice_kernel.hpp), W4, f16, GPT=1 (round 5: the FRAG32 small-batch kernels serve f16 too)
#include <algorithm>

#include "gemm_kslice_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_KSLICE_LAUNCH_SET(4, DIHIP_F16, 1)
}  // namespace dihip
