// explicit instantiations: decode fast path (gemv_stream_kernel.hpp), W8, bf16 activations
#include "gemv_stream_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_GEMV_LAUNCH_SET(8, DIHIP_BF16, 0)
}  // namespace dihip
