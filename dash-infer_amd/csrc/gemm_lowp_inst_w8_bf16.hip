// explicit instantiations: A16W8, bf16 activations (all pro/epilogue forms)
#include "gemm_lowp_launch.hpp"
namespace dihip {
DIHIP_DEFINE_GEMM_LAUNCH_SET_STD(8, DIHIP_BF16)
}  // namespace dihip
