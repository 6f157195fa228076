// explicit instantiations: unquantised 16-bit weights (lm_head / dense Gemm), bf16 + f16
#include "gemm_lowp_launch.hpp"
namespace dihip {
DIHIP_DEFINE_GEMM_LAUNCH_SET_DENSE(DIHIP_BF16)
DIHIP_DEFINE_GEMM_LAUNCH_SET_DENSE(DIHIP_F16)
}  // namespace dihip
