// explicit instantiations: A16W4, f16 activations, fused decode-step forms (general kernel: M > 4 and odd shapes)
#include "gemm_lowp_launch.hpp"
namespace dihip {
DIHIP_DEFINE_GEMM_LAUNCH_SET_FUSED(4, DIHIP_F16)
}  // namespace dihip
