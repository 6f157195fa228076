// explicit instantiations: panel GEMM for batched decode (gemm_panel_kernel.hpp), W4, f16, GPT=0
#include <algorithm>

#include "gemm_panel_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_PANEL_LAUNCH_SET(4, DIHIP_F16, 0)
}  // namespace dihip
