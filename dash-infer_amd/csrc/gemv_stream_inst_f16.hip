// explicit instantiations: decode fast path for f16 activations (op-boundary form: plain prologue,
// standard epilogue), W4 / W8, with and without one-group-per-k-tile
#include "gemv_stream_kernel.hpp"
namespace dihip {
DIHIP_DEFINE_GEMV_LAUNCH(4, DIHIP_F16, 1, PRO_PLAIN, EPI_STD, 0)
DIHIP_DEFINE_GEMV_LAUNCH(4, DIHIP_F16, 4, PRO_PLAIN, EPI_STD, 0)
DIHIP_DEFINE_GEMV_LAUNCH(4, DIHIP_F16, 1, PRO_PLAIN, EPI_STD, 1)
DIHIP_DEFINE_GEMV_LAUNCH(4, DIHIP_F16, 4, PRO_PLAIN, EPI_STD, 1)
DIHIP_DEFINE_GEMV_LAUNCH(8, DIHIP_F16, 1, PRO_PLAIN, EPI_STD, 0)
DIHIP_DEFINE_GEMV_LAUNCH(8, DIHIP_F16, 4, PRO_PLAIN, EPI_STD, 0)
DIHIP_DEFINE_GEMV_LAUNCH(8, DIHIP_F16, 1, PRO_PLAIN, EPI_STD, 1)
DIHIP_DEFINE_GEMV_LAUNCH(8, DIHIP_F16, 4, PRO_PLAIN, EPI_STD, 1)
}  // namespace dihip
