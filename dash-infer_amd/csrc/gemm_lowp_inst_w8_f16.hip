// explicit instantiations: A16W8, f16 activations (operator-level form)
#include "gemm_lowp_launch.hpp"
namespace dihip {
DIHIP_DEFINE_GEMM_LAUNCH_SET_STD(8, DIHIP_F16)
}  // namespace dihip
