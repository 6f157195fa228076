// span_attn_op_hip.cpp -- registers op types "DecOptMHA" / "DecOptMQA" on DeviceType::HIP (class: span_attn_op_hip.h).
#include "span_attn_op_hip.h"

namespace allspark {

REGISTER_OP(DecOptMHA, HIP, SpanAttnOpHIP)
REGISTER_OP(DecOptMQA, HIP, SpanAttnOpHIP)

}  // namespace allspark
