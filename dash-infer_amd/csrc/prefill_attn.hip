// prefill_attn.hip -- causal GQA prefill attention (include/dashinfer_hip.h section 4).
// PLACEHOLDER until the MFMA kernel lands: reports ALLSPARK_INVALID_CALL_ERROR.
#include "device_utils.h"

using namespace dihip;

extern "C" int dihip_prefill_attn(void* stream, void* out, const void* q, const void* k, const void* v, int seq_q,
                                  int seq_k, int q_stride, int kv_stride, int n_heads, int n_groups, int head_size,
                                  int causal, float alpha, int dtype) {
  (void)stream; (void)out; (void)q; (void)k; (void)v; (void)seq_q; (void)seq_k; (void)q_stride; (void)kv_stride;
  (void)n_heads; (void)n_groups; (void)head_size; (void)causal; (void)alpha; (void)dtype;
  set_last_error("dihip_prefill_attn: not implemented yet");
  return DIHIP_INVALID_CALL_ERROR;
}
