// explicit instantiations: panel GEMM for batched decode (gemm_panel_kernel.hpp), W8, f16, GPT=1
#include <algorithm>

#include "gemm_panel_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_PANEL_LAUNCH_SET(8, DIHIP_F16, 1)
}  // namespace dihip
