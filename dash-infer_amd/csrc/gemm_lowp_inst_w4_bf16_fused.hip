// explicit instantiations: A16W4, bf16 activations, fused decode-step forms
#include "gemm_lowp_launch.hpp"
namespace dihip {
DIHIP_DEFINE_GEMM_LAUNCH_SET_FUSED(4, DIHIP_BF16)
}  // namespace dihip
