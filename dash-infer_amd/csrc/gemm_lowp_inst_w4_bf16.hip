// explicit instantiations: A16W4, bf16 activations (all pro/epilogue forms)
#include "gemm_lowp_launch.hpp"
namespace dihip {
DIHIP_DEFINE_GEMM_LAUNCH_SET_STD(4, DIHIP_BF16)
}  // namespace dihip
