// explicit instantiations: A16W8, f16 activations, fused decode-step forms (general kernel: M > 4 and odd shapes)
#include "gemm_lowp_launch.hpp"
namespace dihip {
DIHIP_DEFINE_GEMM_LAUNCH_SET_FUSED(8, DIHIP_F16)
}  // namespace dihip
