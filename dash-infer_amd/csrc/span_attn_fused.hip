// span_attn_fused.hip -- decode-step attention with Rotary and the KV-cache append folded in
// (include/dashinfer_hip.h section 3b).
//
// The reference runs Rotary -> DecOptMQA (cache append kernel, then QK / softmax / PV / reduce
// kernels) per layer (python/pyhie/allspark/model/qwen_v15.py:228-262, span_attn_op.cpp:90-169).
// At batch 1 every one of those launches is latency, not bandwidth.  One entry point does the whole
// sub-graph from the fused qkv row the GEMV just produced (cos / sin from a table built once with the
// same arithmetic as dihip_rope_qk):
//
//   16-bit cache      span_attn_ft_mfma_kernel<.., FUSED> (span_attn_ft_mfma.hpp): Rotary, append and both contractions on
//                     the matrix cores in one launch, + the split merge.
//   int8 cache        the same kernel, MODE = I8 (round 5): every wave of the workgroup that holds the new token quantises this step's
//                     K / V head into its own LDS row as the append kernel does and substitutes codes + parameters in its tiles;
//                     bit-identical to the two launches below (output and span bytes).
//   uint4 cache       bf16 rows: span_attn_u4_mfma_kernel<FUSED> (round 4), one launch; f16 rows: rope_kv_append_kernel
//                     (span_cache.hip: Rotary, quantising append, rotated q into the workspace), then the matrix-core decode
//                     kernels of the op boundary (span_attn.hip) on lengths + 1, + the split merge.
//
// (Until round 2 the quantised caches, and the 16-bit cache before its matrix-core form existed, went through a VALU
// kernel with one query head per wave that did all of it in one launch: 22 - 325 spilled VGPRs depending on the heads per
// group (profiles/r01j_kernel_resources.txt).  It is gone; the append launch costs one kernel boundary.)
//
// No inter-workgroup communication, no host work per step: sequence lengths are read on the device.
#include <algorithm>
#include <cstdlib>

#include "span_attn_common.hpp"

namespace dihip {

// cos / sin per (position, dim pair): the table dihip_span_attn_decode_fused reads (same arithmetic as
// dihip_rope_qk: rope_sincos)
__global__ void rope_table_kernel(float* tab, const float* inv_freq, int max_pos, int half) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= max_pos * half) return;
  const int pos = idx / half, i = idx - pos * half;
  float sn, cs;
  rope_sincos((uint32_t)pos, inv_freq[i], &sn, &cs);
  tab[(size_t)idx * 2] = cs;
  tab[(size_t)idx * 2 + 1] = sn;
}

static size_t fused_q_bytes(int batch, int n_heads) { return ((size_t)batch * n_heads * 128 * 2 + 255) & ~(size_t)255; }

}  // namespace dihip

using namespace dihip;

extern "C" {

int dihip_rope_table(void* stream, float* table, const float* inv_freq, int max_pos, int head_size) {
  DIHIP_REQUIRE(table && inv_freq && max_pos > 0 && head_size > 0 && head_size % 2 == 0, DIHIP_PARAM_ERROR,
                "rope_table: bad argument");
  const int half = head_size / 2;
  const int total = max_pos * half;
  hipLaunchKernelGGL(rope_table_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table,
                     inv_freq, max_pos, half);
  return launch_status();
}

size_t dihip_span_attn_fused_workspace_bytes(int batch, int n_heads, int n_groups, int head_size, int max_seq_len) {
  if (batch <= 0 || n_heads <= 0 || n_groups <= 0 || max_seq_len <= 0 || n_heads % n_groups) return 0;
  (void)head_size;
  // the entry has no cache mode: the largest of the three forms (quantised caches: rotated q + the decode kernels' partials)
  size_t quant = 0;
  for (int mode : {DIHIP_KV_I8, DIHIP_KV_U4})
    quant = std::max(quant, span_attn_decode_workspace_bytes(batch, n_heads, n_groups, max_seq_len, mode, DIHIP_BF16));
  return std::max(fused_q_bytes(batch, n_heads) + quant, span_attn_fused_mfma_workspace_bytes(batch, n_heads, n_groups, max_seq_len)) + 256;
}

int dihip_span_attn_decode_fused(void* stream, void* output, const void* qkv, void* const* k_span_array,
                                 void* const* v_span_array, const uint32_t* old_seq_lens_dev, const float* rope_table,
                                 int batch, int n_heads, int n_groups, int head_size, int span_len,
                                 int n_spans_per_request, int max_seq_len, int kv_mode, int dtype, float qk_scale, void* ws,
                                 size_t ws_bytes) {
  return dihip_span_attn_decode_fused_sync(stream, output, qkv, k_span_array, v_span_array, old_seq_lens_dev, rope_table, batch,
                                           n_heads, n_groups, head_size, span_len, n_spans_per_request, max_seq_len, kv_mode, dtype,
                                           qk_scale, ws, ws_bytes, nullptr, 0);
}

int dihip_span_attn_decode_fused_sync(void* stream, void* output, const void* qkv, void* const* k_span_array,
                                      void* const* v_span_array, const uint32_t* old_seq_lens_dev, const float* rope_table,
                                      int batch, int n_heads, int n_groups, int head_size, int span_len,
                                      int n_spans_per_request, int max_seq_len, int kv_mode, int dtype, float qk_scale, void* ws,
                                      size_t ws_bytes, void* sync, size_t sync_bytes) {
  return dihip_span_attn_decode_step(stream, output, qkv, k_span_array, v_span_array, old_seq_lens_dev, rope_table, batch, n_heads,
                                     n_groups, head_size, span_len, n_spans_per_request, max_seq_len, kv_mode, dtype, qk_scale, ws,
                                     ws_bytes, sync, sync_bytes, DIHIP_ACT_ROWMAJOR);
}

int dihip_span_attn_decode_step(void* stream, void* output, const void* qkv, void* const* k_span_array,
                                void* const* v_span_array, const uint32_t* old_seq_lens_dev, const float* rope_table, int batch,
                                int n_heads, int n_groups, int head_size, int span_len, int n_spans_per_request, int max_seq_len,
                                int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes, void* sync, size_t sync_bytes,
                                int out_layout) {
  DIHIP_REQUIRE(batch >= 0 && n_heads > 0 && n_groups > 0 && n_spans_per_request > 0 && max_seq_len > 0, DIHIP_PARAM_ERROR,
                "span_attn_decode_fused: invalid parameter");
  DIHIP_REQUIRE(output && qkv && k_span_array && v_span_array && old_seq_lens_dev && rope_table, DIHIP_PARAM_ERROR,
                "span_attn_decode_fused: null pointer");
  DIHIP_REQUIRE(head_size == 128, DIHIP_PARAM_ERROR, "span_attn: unsupported head size %d (only 128, dispatch.hpp:45-57)",
                head_size);
  DIHIP_REQUIRE(n_heads % n_groups == 0 && n_heads / n_groups <= 32, DIHIP_PARAM_ERROR,
                "span_attn: nHeads/nGroups must be an integer <= 32 (got %d/%d)", n_heads, n_groups);
  DIHIP_REQUIRE(span_len == 16 || span_len == 32 || span_len == 64 || span_len == 128, DIHIP_PARAM_ERROR,
                "span_attn: span length %d not in {16,32,64,128}", span_len);
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "span_attn_decode_fused: 16-bit activations only");
  DIHIP_REQUIRE(kv_mode == DIHIP_KV_NONE || kv_mode == DIHIP_KV_I8 || kv_mode == DIHIP_KV_U4, DIHIP_PARAM_ERROR,
                "span_attn_decode_fused: unsupported kv mode %d", kv_mode);
  DIHIP_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0, DIHIP_PARAM_ERROR, "span_attn_decode_fused: qkv must be 16-byte aligned");
  DIHIP_REQUIRE(out_layout == DIHIP_ACT_ROWMAJOR || out_layout == DIHIP_ACT_FRAG32, DIHIP_PARAM_ERROR, "span_attn_decode_step: bad out_layout");
  if (batch == 0) return DIHIP_SUCCESS;
  if (kv_mode == DIHIP_KV_U4 || kv_mode == DIHIP_KV_I8) {
    // uint4 cache with bf16 activations (span_attn_u4_mfma_kernel<FUSED>) and int8 cache (span_attn_ft_mfma_kernel<FT, I8, FUSED>,
    // round 5): one launch as well; otherwise (uint4 with f16 rows, DIHIP_ATTN_I8_FUSED=0) the two launches below
    static const bool two_launches = env_off("DIHIP_ATTN_U4_FUSED");  // =0: the append launch + the op-boundary kernel (A/B)
    if (kv_mode == DIHIP_KV_I8 || !two_launches) {
      bool handled = false;
      const int st = span_attn_fused_mfma(stream, output, qkv, k_span_array, v_span_array, old_seq_lens_dev, rope_table, batch,
                                          n_heads, n_groups, span_len, n_spans_per_request, max_seq_len, kv_mode, dtype, qk_scale,
                                          ws, ws_bytes, &handled, sync, sync_bytes, out_layout);
      if (handled) return st;
    }
  }
  DIHIP_REQUIRE(out_layout == DIHIP_ACT_ROWMAJOR, DIHIP_PARAM_ERROR,
                "span_attn_decode_step: FRAG32 output is served by the one-launch uint4 / int8 forms only (batch <= 32)");
  if (kv_mode == DIHIP_KV_NONE) {
    // 16-bit cache: one launch, both contractions on the matrix cores (span_attn.hip); 10.7 vs 15.5 us per layer at batch 1
    bool handled = false;
    const int st = span_attn_fused_mfma(stream, output, qkv, k_span_array, v_span_array, old_seq_lens_dev, rope_table, batch,
                                        n_heads, n_groups, span_len, n_spans_per_request, max_seq_len, kv_mode, dtype, qk_scale,
                                        ws, ws_bytes, &handled, sync, sync_bytes);
    DIHIP_REQUIRE(handled, DIHIP_MEMORY_ERROR, "span_attn_decode_fused: workspace too small (%zu bytes; dihip_span_attn_fused_workspace_bytes)",
                  ws_bytes);
    return st;
  }
  // quantised cache: Rotary + quantising append (rotated q into the workspace), then the decode kernels on lengths + 1
  const size_t qb = fused_q_bytes(batch, n_heads);
  const size_t need = qb + span_attn_decode_workspace_bytes(batch, n_heads, n_groups, max_seq_len, kv_mode, dtype);
  DIHIP_REQUIRE(ws && ws_bytes >= need, DIHIP_MEMORY_ERROR, "span_attn_decode_fused: workspace too small (%zu < %zu)", ws_bytes, need);
  int st = rope_table_kv_append(stream, k_span_array, v_span_array, ws, qkv, old_seq_lens_dev, rope_table, batch, n_heads, n_groups,
                                span_len, n_spans_per_request, kv_mode, dtype);
  if (st != DIHIP_SUCCESS) return st;
  return span_attn_decode_biased(stream, output, ws, k_span_array, v_span_array, old_seq_lens_dev, 1, batch, n_heads, n_groups,
                                 span_len, n_spans_per_request, max_seq_len, kv_mode, dtype, qk_scale,
                                 reinterpret_cast<unsigned char*>(ws) + qb, ws_bytes - qb);
}

}  // extern "C"
