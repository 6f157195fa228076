// explicit instantiations: decode fast path, W8, bf16, one quantisation group per k-tile
#include "gemv_stream_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_GEMV_LAUNCH_SET(8, DIHIP_BF16, 1)
}  // namespace dihip
