// oracle/ref_cuda_shim stand-in for span-attention/src/common/data_type.h (CUTLASS half / bfloat16 aliases: not needed by the
// codec arithmetic, which runs in float).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <span_attn.h>
