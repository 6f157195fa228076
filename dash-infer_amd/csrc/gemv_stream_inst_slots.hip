// explicit instantiations: expert-slot form of the decode GEMV (mixture-of-experts), bf16; MR = 1: one slot per workgroup
// row, MR = 4: groups of up to 4 slots that picked the same expert
#include "gemv_stream_kernel.hpp"
namespace dihip {
#define SLOTS(MR)                                              \
  DIHIP_DEFINE_GEMV_SLOT_LAUNCH(8, DIHIP_BF16, EPI_SWIGLU, 0, MR) \
  DIHIP_DEFINE_GEMV_SLOT_LAUNCH(8, DIHIP_BF16, EPI_STD, 0, MR)    \
  DIHIP_DEFINE_GEMV_SLOT_LAUNCH(8, DIHIP_BF16, EPI_SWIGLU, 1, MR) \
  DIHIP_DEFINE_GEMV_SLOT_LAUNCH(8, DIHIP_BF16, EPI_STD, 1, MR)    \
  DIHIP_DEFINE_GEMV_SLOT_LAUNCH(4, DIHIP_BF16, EPI_SWIGLU, 0, MR) \
  DIHIP_DEFINE_GEMV_SLOT_LAUNCH(4, DIHIP_BF16, EPI_STD, 0, MR)    \
  DIHIP_DEFINE_GEMV_SLOT_LAUNCH(4, DIHIP_BF16, EPI_SWIGLU, 1, MR) \
  DIHIP_DEFINE_GEMV_SLOT_LAUNCH(4, DIHIP_BF16, EPI_STD, 1, MR)
SLOTS(1)
}  // namespace dihip
