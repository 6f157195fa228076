// oracle/ref_cuda_shim stand-in for span-attention/src/common/fp_math.cuh (cuda_fp16 / cuda_bf16 helpers: the codec is compiled for
// T = float here, fed values exactly representable in the product's FT).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "common/func_modifier.h"
