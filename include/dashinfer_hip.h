/* dashinfer_hip.h -- C-ABI of the MI355X (gfx950) device backend for DashInfer's quantized
 * decode hot path.  This is the drop-in boundary: the host-side C++ operators
 * (dash-infer_amd/host, mirroring allspark::AsOperator) and any foreign-language binding call
 * ONLY these functions.  Plain pointers and sizes, no C++ / torch types, no exceptions.
 *
 * Conventions
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every call only
 *     ENQUEUES work on that stream (graph-capturable: no allocation, no synchronisation) unless
 *     stated otherwise.
 *   - all data pointers are DEVICE pointers unless the name ends in `_host`.
 *   - return value: an allspark::AsStatus value (csrc/interface/allspark_check.h:62-80) for
 *     dihip_* operator entry points; a span::SaStatus value
 *     (span-attention/include/spanattn/span_attn.h:50-66) for dihip_span_attn_* (the reference's
 *     inner library boundary keeps its own status enum).
 *   - `dtype` is the activation type "FT": DIHIP_F32 / DIHIP_F16 / DIHIP_BF16, numbered like
 *     span::DataType (span_attn.h:27-34).
 */
#ifndef DASHINFER_HIP_H_
#define DASHINFER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status: allspark::AsStatus (csrc/interface/allspark_check.h:62-80) ------------------- */
#define DIHIP_SUCCESS 0
#define DIHIP_UNKNOWN_ERROR 1
#define DIHIP_PARAM_ERROR 2
#define DIHIP_MEMORY_ERROR 4
#define DIHIP_RUNTIME_ERROR 5
#define DIHIP_EXCEED_LIMIT_ERROR 7
#define DIHIP_INVALID_CALL_ERROR 8

/* ---- status: span::SaStatus (span_attn.h:50-66) ------------------------------------------- */
#define DIHIP_SA_SUCCESS 0
#define DIHIP_SA_HIP_ERROR 1 /* CUDA_ERROR in the reference */
#define DIHIP_SA_RUNTIME_ERROR 2
#define DIHIP_SA_PARAM_ERROR 3
#define DIHIP_SA_EXCEED_LIMIT_ERROR 4
#define DIHIP_SA_INTERNAL_ERROR 5
#define DIHIP_SA_UNKNOWN_ERROR 127

/* ---- enums --------------------------------------------------------------------------------- */
enum { DIHIP_F32 = 0, DIHIP_F16 = 1, DIHIP_BF16 = 2 };          /* span::DataType */
enum { DIHIP_KV_NONE = 0, DIHIP_KV_I8 = 1, DIHIP_KV_U4 = 2 };   /* span::QuantMode */
/* activation = allspark UnaryType (csrc/proto/allspark.proto:68-76) */
enum {
  DIHIP_ACT_NONE = 0, DIHIP_ACT_TANH = 1, DIHIP_ACT_GELU_ERF = 2, DIHIP_ACT_GELU_TANH = 3,
  DIHIP_ACT_RELU = 4, DIHIP_ACT_SILU = 5, DIHIP_ACT_SIGMOID = 6
};

const char* dihip_version(void);
/* last error text of the calling thread (never NULL) */
const char* dihip_last_error(void);
/* number of CUs / name of the current device; returns DIHIP_RUNTIME_ERROR without a GPU */
int dihip_device_info(int* num_cus, int* lds_bytes_per_cu, char* name, size_t name_len);

/* =============================================================================================
 * 1. Weight-only GEMM / GEMV  (replaces GemmA16W8GPU / GemmA16W4GPU and their launchers:
 *    csrc/core/operator/general/gemm_lowp/gemm_a16w8_gpu.cpp:30-267, gemm_a16w4_gpu.cpp:26-230,
 *    csrc/core/kernel/cuda/gemm_lowp/gemm_a16w8_kernel.h:232-529, gemm_a16w4_kernel.h:53-273)
 *
 *    Y[M,N] = act(alpha * X[M,K] . ((Wq[K,N] - Z[G,N]) (.) S[G,N]) + bias[N]) (+ residual[M,N])
 *
 *    Like the reference (N32K16 reorder + u8 bias at InitV2, gemm_a16w8_gpu.cpp:421-473) the
 *    weights are re-laid-out once at init into an MFMA-fragment-major tile order
 *    ("dihip tile-major") by dihip_gemm_lowp_pack(); the GEMM entry points
 *    take the packed buffers.
 * ========================================================================================== */

/* bytes of the packed weight / packed (scale,zero) buffers for a [K,N] weight */
size_t dihip_gemm_lowp_packed_weight_bytes(int wbits, int N, int K);
size_t dihip_gemm_lowp_packed_sz_bytes(int N, int K, int group_size);

/* Re-layout on the device (InitV2-time; enqueued on `stream`).
 *   wbits 8: wq = int8  [K, N]          row-major   (quantization_utils.py:158-217 output)
 *   wbits 4: wq = uint8 [K, ceil(N/2)]  lo nibble = even n (gemm_a16w4.h:25-33)
 *   scales/zeros: FT [G, N], G = 1 (group_size <= 0, per-channel) or ceil(K/group_size)
 *   group_size: -1 or a multiple of 32 (the reference GPU kernels want %32 as well).
 * w_packed / sz_packed: outputs of the sizes returned above.                                  */
int dihip_gemm_lowp_pack(void* stream, int wbits, const void* wq, const void* scales,
                         const void* zeros, int N, int K, int group_size, int dtype,
                         void* w_packed, void* sz_packed);

/* Context phase (M >= 64): when the grid of 128 x 256 tiles would end in a round that fills at most half the chip, the column blocks of
 * that round are split in K (whole quantisation groups per part, parts added in order by a reduction launch): the number of parts per
 * tail tile for this shape on this GPU, 1 = no split.  Diagnostics / tests; DIHIP_PREFILL_TAIL_SPLIT=0 turns the split off. */
int dihip_gemm_prefill_tail_parts(int wbits, int M, int N, int K, int group_size, int dual);
/* scratch (split-K slabs; contents need no initialisation) and sync (arrival counters; must be
 * zero-filled ONCE by the owner, the kernels leave it zero) sizes */
size_t dihip_gemm_lowp_workspace_bytes(int wbits, int M, int N, int K, int group_size);
size_t dihip_gemm_lowp_sync_bytes(void);

/* op type "GemmA16W8" / "GemmA16W4".  x: FT [M, K] (row stride K); y: FT [M, N].
 * bias: FT [N] or NULL; residual: FT [M, N] added after the activation, or NULL (the Gemm op's
 * fused binary ADD, python/pyhie/allspark/model/qwen_v15.py:296-300); act: DIHIP_ACT_*.
 * sync: >= dihip_gemm_lowp_sync_bytes() zero-initialised device bytes owned by the op, or NULL
 * (then the arrival counters live in `ws` and are cleared with a memset node per call).       */
int dihip_gemm_a16w8(void* stream, const void* x, const void* w_packed, const void* sz_packed,
                     const void* bias, const void* residual, void* y, int M, int N, int K,
                     int group_size, int act, float alpha, void* ws, size_t ws_bytes, void* sync,
                     int dtype);
int dihip_gemm_a16w4(void* stream, const void* x, const void* w_packed, const void* sz_packed,
                     const void* bias, const void* residual, void* y, int M, int N, int K,
                     int group_size, int act, float alpha, void* ws, size_t ws_bytes, void* sync,
                     int dtype);

/* Fused decode-step variants (the glue the reference runs as separate
 * LayerNormNoBeta / Binary / Unary ops, python/pyhie/allspark/model/qwen_v15.py:210-381).
 * hidden stream `h` is f32 [M, K]; FT = bf16 or f16 (dtype; the FRAG32 layouts and the small-batch kernels behind them are bf16).
 *   norm_gemm   : y = act(rmsnorm(h; gamma, eps) . W + bias)                 y: FT [M,N]
 *   norm_swiglu : y = FT(silu(rmsnorm(h).Wg)) * FT(rmsnorm(h).Wu)            y: FT [M,N]
 *   gemm_addto  : h_out[M,N] (f32) = h_res + x . W   (f32, no rounding)           x: FT [M,K]   */
int dihip_fused_norm_gemm(void* stream, int wbits, const float* h, const void* gamma, float eps,
                          const void* w_packed, const void* sz_packed, const void* bias, void* y,
                          int M, int N, int K, int group_size, int act, void* ws, size_t ws_bytes,
                          void* sync, int dtype);
int dihip_fused_norm_swiglu(void* stream, int wbits, const float* h, const void* gamma, float eps,
                            const void* wg_packed, const void* szg_packed, const void* wu_packed,
                            const void* szu_packed, void* y, int M, int N, int K, int group_size,
                            void* ws, size_t ws_bytes, void* sync, int dtype);
int dihip_fused_gemm_addto(void* stream, int wbits, const void* x, const void* w_packed,
                           const void* sz_packed, const float* h_res, float* h_out, int M, int N,
                           int K, int group_size, void* ws, size_t ws_bytes, void* sync, int dtype);

/* Activation layouts between two calls of the decode-step section.  DIHIP_ACT_FRAG32 stores the
 * 16-bit matrix x[M <= 32, K] as the MFMA A fragments the small-batch kernel (1 < M <= 32) consumes:
 * element (m, k) at  ((((k/32)*MT + m/16)*64 + ((k%32)/8)*16 + m%16)*8 + k%8,  MT = 1 (M <= 16) or 2;
 * a fragment is then one contiguous 1 KiB wave-load instead of 16 pieces of 64 B (1.2-1.7x faster
 * kernels).  dihip_gemm_lowp_prefers_frag() tells whether a [M, N, K] call (dual = 1: the
 * gate/up pair of dihip_fused_norm_swiglu) runs on that kernel; only then may FRAG32 be passed.   */
#define DIHIP_ACT_ROWMAJOR 0
#define DIHIP_ACT_FRAG32 1
int dihip_gemm_lowp_prefers_frag(int wbits, int M, int N, int K, int group_size, int dual);
size_t dihip_act_frag_bytes(int M, int K);
int dihip_act_to_frag(void* stream, const void* x_rowmajor, void* x_frag, int M, int K, int dtype);
int dihip_act_from_frag(void* stream, const void* x_frag, void* x_rowmajor, int M, int K, int dtype);
int dihip_fused_norm_swiglu_ex(void* stream, int wbits, const float* h, const void* gamma, float eps,
                               const void* wg_packed, const void* szg_packed, const void* wu_packed,
                               const void* szu_packed, void* y, int M, int N, int K, int group_size,
                               void* ws, size_t ws_bytes, void* sync, int dtype, int y_layout);
int dihip_fused_gemm_addto_ex(void* stream, int wbits, const void* x, const void* w_packed,
                              const void* sz_packed, const float* h_res, float* h_out, int M, int N,
                              int K, int group_size, void* ws, size_t ws_bytes, void* sync, int dtype,
                              int x_layout);
/* Batched decode (M > 4), fewer launches: the RMSNorm that FOLLOWS a residual GEMM in the decoder graph
 * (LayerNormNoBeta after o_proj / down_proj, qwen_v15.py:296-361) is produced by the GEMM call itself --
 *   h_out = h_res + x.W;   xnorm = RMSNorm(h_out; gamma, eps)  in FT, row-major [M, N] or FRAG32
 * -- riding on the split-K slab reduction when the plan has one (one workgroup per row sums the slices,
 * adds the residual and normalises: the separate norm launch and one pass over the row disappear), as a
 * separate launch otherwise; the result is the same either way and equal to dihip_fused_gemm_addto_ex followed
 * by the norm of dihip_fused_norm_gemm / _swiglu.  The next GEMMs then take the normalised rows directly:
 * dihip_prenorm_gemm (qkv: + bias) and dihip_prenorm_swiglu (gate / up).  x_layout / xnorm_layout as above. */
int dihip_fused_gemm_addto_norm(void* stream, int wbits, const void* x, const void* w_packed,
                                const void* sz_packed, const float* h_res, float* h_out, int M, int N,
                                int K, int group_size, void* ws, size_t ws_bytes, void* sync, int dtype,
                                int x_layout, const void* gamma, float eps, void* xnorm, int xnorm_layout);
/* LayerNormNoBeta (csrc/core/kernel/cpu/layernorm.cpp:110-157) of M rows of the f32 hidden stream into FT rows (row-major),
 * as its own call: the norm every fused entry above applies, for graphs with several consumers of it (the MoE layer) */
int dihip_rmsnorm_rows(void* stream, void* xnorm, const float* h, const void* gamma, float eps, int M, int K,
                       int dtype);
int dihip_prenorm_gemm(void* stream, int wbits, const void* xnorm, int x_layout, const void* w_packed,
                       const void* sz_packed, const void* bias, void* y, int M, int N, int K,
                       int group_size, int act, void* ws, size_t ws_bytes, void* sync, int dtype);
int dihip_prenorm_swiglu(void* stream, int wbits, const void* xnorm, int x_layout, const void* wg_packed,
                         const void* szg_packed, const void* wu_packed, const void* szu_packed, void* y,
                         int M, int N, int K, int group_size, void* ws, size_t ws_bytes, void* sync,
                         int dtype, int y_layout);
/* ... with the RMSNorm DEFERRED (round 5): 1 / rms of a row is a scalar and commutes with the GEMM, so the LayerNormNoBeta between a
 * residual GEMM and the next GEMM (qwen_v15.py:296-361) needs neither a launch nor a pass of its own --
 *   producer  dihip_fused_gemm_addto_prenorm:  h_out = h_res + x.W;  xnorm = FT(gamma * h_out) (NO 1 / rms);
 *             rowsq[part][32] = partial sums of h_out[m][.]^2, one part per workgroup, *rowsq_parts of them (fixed order);
 *             *rowsq_parts == 0: the serving kernel does not offer it and xnorm is the FINISHED norm (dihip_fused_gemm_addto_norm);
 *   consumer  dihip_prenorm_gemm_rowsq / dihip_prenorm_swiglu_rowsq:  every accumulator of row m is multiplied by
 *             1 / sqrt(Sum_p rowsq[p][m] / K + eps) before alpha / bias / SwiGLU (rowsq_parts == 0: plain dihip_prenorm_*).
 * The reference rounds (gamma * x) * rstd to FT; here gamma * x is rounded and rstd is applied to the f32 accumulator: the
 * same relative precision, NOT the same bits (tests: against the oracle's tolerance, and the full-depth parity cases).
 * bf16, 4 < M <= 32.  dihip_prenorm_rowsq_supported(...): does the kernel that will serve the consumer take deferred rows
 * (dual = 1: the SwiGLU pair)?  Ask BEFORE calling the producer; rowsq: dihip_rowsq_bytes() bytes, 16-byte aligned.
 * DIHIP_DEFER_RMSNORM=0: never supported (A/B). */
int dihip_fused_gemm_addto_prenorm(void* stream, int wbits, const void* x, const void* w_packed,
                                   const void* sz_packed, const float* h_res, float* h_out, int M, int N,
                                   int K, int group_size, void* ws, size_t ws_bytes, void* sync, int dtype,
                                   int x_layout, const void* gamma, float eps, void* xnorm, int xnorm_layout,
                                   float* rowsq, size_t rowsq_bytes, int* rowsq_parts);
int dihip_prenorm_gemm_rowsq(void* stream, int wbits, const void* xnorm, int x_layout, const void* w_packed,
                             const void* sz_packed, const void* bias, void* y, int M, int N, int K,
                             int group_size, int act, void* ws, size_t ws_bytes, void* sync, int dtype,
                             const float* rowsq, int rowsq_parts, float eps);
int dihip_prenorm_swiglu_rowsq(void* stream, int wbits, const void* xnorm, int x_layout, const void* wg_packed,
                               const void* szg_packed, const void* wu_packed, const void* szu_packed, void* y,
                               int M, int N, int K, int group_size, void* ws, size_t ws_bytes, void* sync,
                               int dtype, int y_layout, const float* rowsq, int rowsq_parts, float eps);
int dihip_prenorm_rowsq_supported(int wbits, int M, int N, int K, int group_size, int dual, int dtype, int x_layout);
size_t dihip_rowsq_bytes(void);

/* ---------------------------------------------------------------------------------------------
 * 1b. Mixture-of-experts decode path with weight-only experts (BASELINE configs[4]).
 *     Semantics of the reference's MOE operator (csrc/core/operator/general/moe/moe_op.cpp:338-460): float
 *     softmax of the router logits, top-k WITHOUT renormalisation, per (token, expert) SiLU(x.Wgate) * (x.Wup)
 *     -> .Wdown, out[t] = sum_k score[t,k] * y[t,k] in float.  The reference's experts are bf16 / A8W8; these
 *     are A16W8 / A16W4 (section 1 quantiser and packing), one packed tensor per expert, experts stacked back
 *     to back: expert e of a projection lives at  base + e * dihip_gemm_lowp_packed_weight_bytes(wbits, N, K)
 *     (scales/zeros: + e * dihip_gemm_lowp_packed_sz_bytes(N, K, group)); gate/up: N = proj, K = hidden;
 *     down: N = hidden, K = proj.  An expert index < 0 in `experts` skips the slot (expert parallelism).
 *   dihip_moe_route  : scores f32 [T, top_k], experts i32 [T, top_k] (descending probability, lower index
 *                      first on ties) from router logits FT/f32 [T, num_experts <= 256]
 *   dihip_moe_experts: out FT [T, hidden]; ws >= dihip_moe_workspace_bytes (no initialisation needed)      */
int dihip_moe_route(void* stream, const void* router_logits, int num_tokens, int num_experts, int top_k,
                    float* scores, int32_t* experts, int dtype);
/* expert parallelism (attribute use_ep, moe_op.cpp:103-117: rank r owns experts [r * E / nranks, (r + 1) * E / nranks)):
 * the indices come out as positions in the rank's own stack [ep_first, ep_first + ep_count), -1 for experts held elsewhere
 * (dihip_moe_experts skips those slots; the all-reduce that follows the operator supplies their terms) */
int dihip_moe_route_ep(void* stream, const void* router_logits, int num_tokens, int num_experts, int top_k,
                       float* scores, int32_t* experts, int dtype, int ep_first, int ep_count);
size_t dihip_moe_workspace_bytes(int num_tokens, int top_k, int hidden, int proj);
int dihip_moe_experts(void* stream, int wbits, const void* x, const int32_t* experts, const float* scores,
                      const void* gate_packed, const void* gate_sz, const void* up_packed,
                      const void* up_sz, const void* down_packed, const void* down_sz, int num_tokens,
                      int top_k, int hidden, int proj, int group_size, void* out, void* ws,
                      size_t ws_bytes, int dtype);
/* CalcExpertKernelLauncher (csrc/core/kernel/cuda/calc_expert.cu:27-35; op CalcExpert): out[t, :] = in[t, :] * expert_weight[t] */
int dihip_calc_expert(void* stream, void* out, const void* in, const void* expert_weight, int num_tokens,
                      int hidden, int dtype);
/* Tail of the MoE layer graph (python/pyhie/allspark/model/qwen_v20_moe.py:366-382): CalcExpert (shared-expert output x its
 * sigmoid gate, csrc/core/kernel/cuda/calc_expert.cu:27-35, rounded to FT) + expert_add + final_add on the f32 hidden rows:
 *   h_out[t, :] = (h_res ? h_res[t, :] : 0) + moe_out[t, :] + FT(shared_out[t, :] * shared_gate[t])
 * moe_out / shared_out FT [T, hidden], shared_gate FT [T] (the Gemm with activation SIGMOID, N = 1); h_res == NULL on ranks
 * that do not carry the residual (the sum over ranks is the all-reduce of h_out that follows).  In place (h_out == h_res) is fine. */
int dihip_moe_shared_combine(void* stream, float* h_out, const float* h_res, const void* moe_out,
                             const void* shared_out, const void* shared_gate, int num_tokens, int hidden,
                             int dtype);
/* The decode block in fewer launches (bit-identical to the calls above; at most 2048 (token, expert) slots, more than one token):
 *   dihip_moe_route_grouped : dihip_moe_route_ep + the (expert, <= 4 slots) grouping dihip_moe_experts would launch, in one
 *                             workgroup; the tables go to `ws` (>= dihip_moe_workspace_bytes) where dihip_moe_experts_ex expects them
 *   dihip_moe_experts_ex    : flags DIHIP_MOE_PREGROUPED (skip the grouping launch), DIHIP_MOE_NO_FINALIZE (leave the slot
 *                             outputs in `ws`; `out` may be NULL)
 *   dihip_moe_combine       : finalize-routing (sum_k score * slot output, rounded to FT) + dihip_moe_shared_combine's tail */
#define DIHIP_MOE_PREGROUPED 1
#define DIHIP_MOE_NO_FINALIZE 2
int dihip_moe_route_grouped(void* stream, const void* router_logits, int num_tokens, int num_experts, int top_k, float* scores,
                            int32_t* experts, int dtype, int ep_first, int ep_count, int hidden, int proj, void* ws, size_t ws_bytes);
int dihip_moe_experts_ex(void* stream, int wbits, const void* x, const int32_t* experts, const float* scores,
                         const void* gate_packed, const void* gate_sz, const void* up_packed, const void* up_sz,
                         const void* down_packed, const void* down_sz, int num_tokens, int top_k, int hidden, int proj,
                         int group_size, void* out, void* ws, size_t ws_bytes, int dtype, int flags);
int dihip_moe_combine(void* stream, float* h_out, const float* h_res, const void* ws, const float* scores, const int32_t* experts,
                      const void* shared_out, const void* shared_gate, int num_tokens, int top_k, int hidden, int proj, int dtype);
/* The layer graph's two unquantised skinny Gemm operators on the normalised rows in one launch (qwen_v20_moe.py:330-338 "mlp.gate",
 * :360-365 "shared_expert_gate" with activation SIGMOID): router_logits FT [T, E] = FT(xn . W_router), shared_gate FT [T] =
 * FT(sigmoid(xn . w_gate)); both weights packed by dihip_dense_pack ([hidden, E] and [hidden, 1]). */
int dihip_moe_router_gate(void* stream, const void* xn, const void* w_router_packed, const void* w_gate_packed, void* router_logits,
                          void* shared_gate, int num_tokens, int num_experts, int hidden, int dtype);

/* =============================================================================================
 * 2. KV span writers (replace csrc/core/kernel/cuda/cuda_kernel_span_cache.h:12-41)
 *    span layout (bit-compatible with the reference, decoder_cache_append.cuh:33-87):
 *      [g][S][H*bits/8] data, then (quantised modes) [g][S]{f32 zero, f32 scale}
 * ========================================================================================== */
size_t dihip_span_bytes(int num_groups, int span_len, int head_size, int kv_mode, int dtype);

/* DecoderCacheAppendLauncher: split the fused qkv rows [B, (n+2g)*H] into q_out [B, n*H] and
 * append this step's K/V head vectors at token position old_seq_lens[b] of request b.
 * k_spans / v_spans: device arrays [B][span_stride] of span pointers.                         */
int dihip_kv_append(void* stream, void* const* k_spans, void* const* v_spans, void* q_out,
                    const void* qkv, const uint32_t* old_seq_lens, int batch, int num_heads,
                    int num_groups, int head_size, int span_len, int span_stride, int kv_mode,
                    int dtype);

/* Rotary + DecoderCacheAppend in one launch (decode-step glue): identical
 * results to dihip_rope_qk (positions = old_seq_lens) followed by dihip_kv_append.  H == 128.    */
int dihip_rope_kv_append(void* stream, void* const* k_spans, void* const* v_spans, void* q_out,
                         const void* qkv, const uint32_t* old_seq_lens, const float* inv_freq,
                         int batch, int num_heads, int num_groups, int head_size, int span_len,
                         int span_stride, int kv_mode, int dtype);

/* ContextSpanCopyLauncher: contiguous prefill K or V [seq_len, g, H] (row stride `src_stride`
 * elements) -> the request's spans, starting at token `start_pos` (a multiple of span_len).   */
int dihip_kv_context_copy(void* stream, void* const* spans, const void* src, int src_stride,
                          int seq_len, int start_pos, int num_groups, int head_size, int span_len,
                          int kv_mode, int dtype);

/* PrefixCacheCopyLauncher: spans -> contiguous dst [prefix_len, g, H] FT with dequantisation.  */
int dihip_kv_prefix_gather(void* stream, void* dst, void* const* spans, int prefix_len,
                           int num_groups, int head_size, int span_len, int kv_mode, int dtype);

/* SpanToContCopyLauncher / ContToSpanCopyLauncher: raw byte gather/scatter of whole spans.     */
int dihip_span_gather(void* stream, void* dst_cont, void* const* spans, int num_spans,
                      size_t span_bytes);
int dihip_span_scatter(void* stream, void* const* spans, const void* src_cont, int num_spans,
                       size_t span_bytes);

/* =============================================================================================
 * 3. SpanAttention decode (replaces span::CreateHandle / Run / ..., span_attn.h:108-175)
 *    Same argument meaning and SaStatus error behaviour as the reference library; the
 *    cudaDeviceProp argument is replaced by the number of CUs (0 = query the current device).
 *    Constraints as the reference: head_size == 128, nHeads % nGroups == 0,
 *    nHeads/nGroups <= 32, span_len in {16,32,64,128}.
 * ========================================================================================== */
typedef struct dihip_span_attn_handle* dihip_span_attn_handle_t;

int dihip_span_attn_create_handle(dihip_span_attn_handle_t* handle, int dtype, int kv_mode,
                                  int batch, int n_heads, int n_groups, int head_size,
                                  int span_len, int n_spans_per_request, const int* seq_len_host,
                                  int num_cus);
int dihip_span_attn_destroy_handle(dihip_span_attn_handle_t handle);
int dihip_span_attn_host_workspace_bytes(size_t* bytes, dihip_span_attn_handle_t handle);
int dihip_span_attn_device_workspace_bytes(size_t* bytes, dihip_span_attn_handle_t handle);
/* output/query: FT [batch, n_heads, head_size]; k/v_span_array: device [batch][n_spans_per_request] */
int dihip_span_attn_run(void* output, const void* query, const void* const* k_span_array,
                        const void* const* v_span_array, void* device_ws, size_t device_ws_bytes,
                        void* host_ws, size_t host_ws_bytes, float qk_scale,
                        dihip_span_attn_handle_t handle, void* stream);

/* Handle-free form for graph replay: sequence lengths (INCLUDING the new token) are read from
 * device memory, so nothing on the host changes between decode steps.  max_seq_len bounds the
 * split count (and the workspace).  Returns an AsStatus value.  Split sequences are combined by a
 * second small launch; `sync` (dihip_span_attn_sync_bytes) is no longer touched and may be NULL -- it
 * stays in the signature for callers written against the earlier in-kernel merge.              */
size_t dihip_span_attn_decode_workspace_bytes(int batch, int n_heads, int head_size,
                                              int max_seq_len, int num_cus);
int dihip_span_attn_decode(void* stream, void* output, const void* query,
                           const void* const* k_span_array, const void* const* v_span_array,
                           const uint32_t* seq_lens_dev, int batch, int n_heads, int n_groups,
                           int head_size, int span_len, int n_spans_per_request, int max_seq_len,
                           int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes,
                           void* sync);
/* same, output optionally in the FRAG32 activation layout (section 1, batch <= 32) for the o-projection
 * that follows (out_layout: DIHIP_ACT_ROWMAJOR / DIHIP_ACT_FRAG32; buffer of dihip_act_frag_bytes(batch, n*H)) */
int dihip_span_attn_decode_ex(void* stream, void* output, const void* query,
                              const void* const* k_span_array, const void* const* v_span_array,
                              const uint32_t* seq_lens_dev, int batch, int n_heads, int n_groups,
                              int head_size, int span_len, int n_spans_per_request, int max_seq_len,
                              int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes,
                              void* sync, int out_layout);
size_t dihip_span_attn_sync_bytes(int batch, int n_heads);
/* The same with the split partials merged INSIDE the launch: `sync` = at least dihip_span_attn_sync_bytes(batch, n_heads) bytes
 * (one 128-byte line of arrival words per request and head) that the caller zeroes ONCE; every call leaves them zeroed; calls
 * sharing one buffer must be ordered on one stream.  sync == NULL or sync_bytes too small: the two-launch merge of
 * dihip_span_attn_decode_ex (never an out-of-bounds ticket).  Bit-identical output either way.                      */
int dihip_span_attn_decode_sync(void* stream, void* output, const void* query,
                                const void* const* k_span_array, const void* const* v_span_array,
                                const uint32_t* seq_lens_dev, int batch, int n_heads, int n_groups,
                                int head_size, int span_len, int n_spans_per_request, int max_seq_len,
                                int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes,
                                void* sync, size_t sync_bytes, int out_layout);

/* 3b. Decode-step form with Rotary and DecoderCacheAppend folded in: replaces the
 * Rotary op (csrc/core/kernel/cpu/rotary.cpp:22-106 semantics, rotate-half, position = old_seq_lens[b])
 * + SpanAttnOp::runDecoder (span_attn_op.cpp:90-169) for one decode step.  Results are identical to
 * dihip_rope_qk + dihip_kv_append + dihip_span_attn_decode: q and k are rotated and rounded to FT, this
 * step's K/V head vectors are written into the spans at token position old_seq_lens[b] (quantised per
 * kv_mode), attention runs over old_seq_lens[b] + 1 tokens.
 *   qkv        : FT [batch, (n + 2g) * H], pre-Rotary fused rows (16-byte aligned)
 *   rope_table : f32 [max_pos][H/2]{cos, sin} built once by dihip_rope_table
 *   max_seq_len: upper bound of old_seq_lens[b] + 1 (sizes the split plan, as in 3)
 * 16-bit cache: one launch (+ split merge); int8 / uint4 cache: the append launch, then the decode kernels of 3.
 *   ws         : >= dihip_span_attn_fused_workspace_bytes(...) (contents need no initialisation)  */
int dihip_rope_table(void* stream, float* table, const float* inv_freq, int max_pos, int head_size);
size_t dihip_span_attn_fused_workspace_bytes(int batch, int n_heads, int n_groups, int head_size,
                                             int max_seq_len);
int dihip_span_attn_decode_fused(void* stream, void* output, const void* qkv, void* const* k_span_array,
                                 void* const* v_span_array, const uint32_t* old_seq_lens_dev,
                                 const float* rope_table, int batch, int n_heads, int n_groups,
                                 int head_size, int span_len, int n_spans_per_request, int max_seq_len,
                                 int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes);
/* The same with the split partials merged INSIDE the launch (16-bit cache, split sequences): `sync` = at least
 * dihip_span_attn_sync_bytes(batch, n_heads) bytes that the caller zeroes ONCE; every call leaves them zeroed (arrival
 * tickets: the workgroup that arrives last for a (request, KV group) merges all split records and writes the output --
 * records stored write-through, read past the L1, no fences).  Bit-identical output; one launch instead of two.
 * sync = NULL (or the quantised caches): as dihip_span_attn_decode_fused.  Calls sharing one `sync` buffer must be
 * ordered on one stream.  Replaces the reference's reduce kernel (span_attention.hpp:145-211).  */
int dihip_span_attn_decode_fused_sync(void* stream, void* output, const void* qkv, void* const* k_span_array,
                                      void* const* v_span_array, const uint32_t* old_seq_lens_dev,
                                      const float* rope_table, int batch, int n_heads, int n_groups,
                                      int head_size, int span_len, int n_spans_per_request, int max_seq_len,
                                      int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes,
                                      void* sync, size_t sync_bytes);
/* The same with the output layout of dihip_span_attn_decode_sync (DIHIP_ACT_ROWMAJOR / DIHIP_ACT_FRAG32).  Round 4: the uint4
 * cache with bf16 activations is ONE launch as well -- Rotary, the quantising DecoderCacheAppend of the new token (byte-identical
 * to dihip_rope_kv_append) and the attention with its split merge -- instead of the append launch + the op-boundary kernel;
 * FRAG32 output is served by the one-launch forms only (batch <= 32).  Round 5: the int8 cache (bf16 / f16 rows) is ONE launch as
 * well (span_attn_ft_mfma_kernel<FT, I8, FUSED>: every wave of the workgroup that holds the new token quantises this step's K / V
 * head into its own LDS row with the arithmetic of the append kernel and substitutes codes + parameters in its tiles; span bytes
 * identical to dihip_rope_kv_append; DIHIP_ATTN_I8_FUSED=0 keeps the append launch).  The runners use it for batch <= 4, where it
 * saves 1.8 us per layer (at batch 32 it measured +1.3: profiles/r05_i8_decode_step.txt).  f16 rows with the uint4 cache keep the
 * append launch.  Workspace / sync as dihip_span_attn_decode_fused_sync. */
int dihip_span_attn_decode_step(void* stream, void* output, const void* qkv, void* const* k_span_array,
                                void* const* v_span_array, const uint32_t* old_seq_lens_dev, const float* rope_table,
                                int batch, int n_heads, int n_groups, int head_size, int span_len,
                                int n_spans_per_request, int max_seq_len, int kv_mode, int dtype, float qk_scale,
                                void* ws, size_t ws_bytes, void* sync, size_t sync_bytes, int out_layout);

/* merge of the per-split partial records  f32 [batch * n_heads][nsplits][132] = { o[128] (unnormalised), m, l, pad }  of a
 * decode attention launch into the FT [batch, n_heads * 128] output (the second launch of 3b) */
int dihip_span_attn_merge_partials(void* stream, void* output, const float* partials, int batch, int n_heads,
                                   int nsplits, int dtype);
/* 3e. The attention half of a batch-1 decode layer as ONE launch (round 5): LayerNormNoBeta -> Gemm[A16W4 | A16W8](qkv, +bias) -> Rotary ->
 * DecOptMQA (DecoderCacheAppend + SpanAttention + reduce) -> Gemm[A16W4 | A16W8](o) -> Binary ADD of the reference graph
 * (qwen_v15.py:210-300; the operator loop it shortens: csrc/core/model/model.cpp:1248-1325).  Equivalent, BIT FOR BIT, to
 *     dihip_fused_norm_gemm(h_in ...) ; dihip_span_attn_decode_fused_sync(...) ; dihip_fused_gemm_addto(attn, o ..., h_res, h_out)
 * at M = 1: the qkv and o weights and the K / V tiles are requested in the first microsecond of the launch, the three operators
 * hand their rows over INSIDE it (8-byte {value, tag} granules, agent-scope stores / loads, bounded waits; csrc/decode_attn_block.hip).
 *   h_in   : f32 [hidden] input of the norm (16-byte aligned);  h_res : f32 [hidden] residual or NULL (row-parallel TP ranks > 0);
 *   h_out  : f32 [hidden] (may alias h_in / h_res);  gamma: bf16 [hidden];  qkv_* / o_*: packed weights (section 1) -- int4 with a group per 128 k
 *            (wbits 4, group_size 128: the headline) and, since round 6, int8 per channel (wbits 8, group_size -1: InstantQuant, BASELINE
 *            configs[1]) and the other forms the decode GEMV streams (int8 g64 / g128, int4 per channel / g256) --, bf16 bias or NULL
 *   ws     : >= dihip_decode_attn_block_workspace_bytes(...) (no initialisation)
 *   sync   : >= dihip_decode_attn_block_sync_bytes(...), 16-byte aligned, zeroed ONCE by the caller; calls sharing it must be ordered
 *            on one stream (hipGraph replay included: the launch keeps its own epoch in it).  Word 1 of `sync` is an error flag: non-zero
 *            after a launch whose bounded wait gave up (results are then undefined, nothing hangs).  The caller READS it at its next
 *            synchronisation point: dihip_decode_attn_block_status_async() enqueues the copy of the word to host memory on the stream
 *            (beside the token readback: no extra synchronisation); non-zero -> the step's results are invalid, dihip_decode_attn_block_reset()
 *            restores the buffer's invariants (epoch, record buffers) and the caller keeps the three-call chain from there
 *            (host/model_runner.cpp Sync; decoder.DecodeSession.check_handoffs).  The wait can only give up when the launch's workgroups
 *            are not all resident (every one of them is needed for the others to finish): _supported() checks the grid against the CU count
 *            AND the kernel's occupancy, but a CU mask or another stream's kernel on the same GPU can still take CUs away.
 *            Since round 6 the split records of the attention live in `sync` too (two buffers alternating by launch: polled by their
 *            consumers, zeroed by them for the launch after next); DIHIP_ATTN_BLOCK_FAULT=1 (tests) makes one workgroup withhold its
 *            rows so that the waits time out.
 * _supported() == 0 (other batch sizes, dtypes -- f16 --, caches, too few CUs, DIHIP_ATTN_BLOCK=0,
 * DIHIP_ATTN_BLOCK_W8=0 for int8): keep the three calls. */
int dihip_decode_attn_block_supported(int wbits, int group_size, int hidden, int n_heads, int n_groups, int head_size,
                                      int max_seq_len, int kv_mode, int dtype, int batch);
size_t dihip_decode_attn_block_sync_bytes(int n_heads, int n_groups, int head_size);
size_t dihip_decode_attn_block_workspace_bytes(int n_heads, int n_groups, int head_size, int max_seq_len);
int dihip_decode_attn_block_status_async(void* stream, const void* sync, unsigned* host_word);
int dihip_decode_attn_block_reset(void* stream, void* sync, size_t sync_bytes);
/* A sync buffer remembers the split plan its polled records were last used with; a launch with ANOTHER plan (max_seq_len moved to another
 * split count) first clears the record region -- on the stream, eagerly.  Inside a stream capture that clear must not happen (a captured
 * memset would run at every replay): a caller that captures its step calls _prepare() with the plan's max_seq_len OUTSIDE the capture
 * whenever the plan may have changed (host/fused_ops_hip.cpp does at Reshape); a launch that finds the plan changed under capture fails with
 * DIHIP_RUNTIME_ERROR.  No-op when nothing changed. */
int dihip_decode_attn_block_prepare(void* stream, void* sync, size_t sync_bytes, int n_heads, int n_groups, int head_size, int max_seq_len);
int dihip_decode_attn_block(void* stream, int wbits, const float* h_in, const float* h_res, float* h_out, const void* gamma,
                            float eps, const void* qkv_w, const void* qkv_sz, const void* qkv_bias, const void* o_w,
                            const void* o_sz, void* const* k_span_array, void* const* v_span_array,
                            const uint32_t* old_seq_lens_dev, const float* rope_table, int hidden, int n_heads, int n_groups,
                            int head_size, int group_size, int span_len, int n_spans_per_request, int max_seq_len, int kv_mode,
                            int dtype, float qk_scale, void* ws, size_t ws_bytes, void* sync, size_t sync_bytes);

/* 3f. The feed-forward half of a batch-1 decode layer as ONE launch (round 5): LayerNormNoBeta -> Gemm[A16W4](gate, SILU) ||
 * Gemm[A16W4](up) -> Binary MUL -> Gemm[A16W4](down) -> Binary ADD (qwen_v15.py:300-388).  Equivalent, BIT FOR BIT, to
 *     dihip_fused_norm_swiglu(h_in ...) ; dihip_fused_gemm_addto(act, down ..., h_res, h_out)
 * at M = 1: the SwiGLU row is handed over inside the launch (granules + one flag word per producing workgroup) while the down
 * projection's first 64 KB per workgroup are already streaming (csrc/decode_mlp_block.hip).  With 3e a decode layer is two launches.
 *   h_in / h_res / h_out as in 3e;  sync: >= dihip_decode_mlp_block_sync_bytes(inter), 16-byte aligned, zeroed ONCE (word 1 = error flag);
 *   calls sharing `sync` must be ordered on one stream (hipGraph replay included).  _supported() == 0: keep the two calls. */
int dihip_decode_mlp_block_supported(int wbits, int group_size, int hidden, int inter, int dtype, int batch);
size_t dihip_decode_mlp_block_sync_bytes(int inter);
int dihip_decode_mlp_block(void* stream, int wbits, const float* h_in, const float* h_res, float* h_out, const void* gamma, float eps,
                           const void* gate_w, const void* gate_sz, const void* up_w, const void* up_sz, const void* down_w,
                           const void* down_sz, int hidden, int inter, int group_size, int dtype, void* sync, size_t sync_bytes);

/* =============================================================================================
 * 4. Prefill attention (replaces xformer_prefill_attention,
 *    csrc/core/kernel/cuda/xformer_mha/xformer_mha.h:26-41): causal softmax(alpha Q K^T) V, GQA.
 *    q: [seq_q, n, H] with row stride q_stride elements; k/v: [seq_k, g, H] with row stride
 *    kv_stride (INTERLEAVED qkv rows: q_stride = kv_stride = (n+2g)*H; MIX: k/v contiguous).
 *    query i attends keys j <= i + (seq_k - seq_q).  out: FT [seq_q, n*H].
 *    head_size 128, FT = bf16 / f16, rows 16-byte aligned; alpha > 0 (the scale is folded into an exp2 argument);
 *    K / V of one call below 2 GiB ((seq_k + 448) * kv_stride * 2 bytes: 32-bit buffer offsets) -> DIHIP_PARAM_ERROR /
 *    DIHIP_EXCEED_LIMIT_ERROR otherwise.  No workspace; seq_q = 0 is a no-op.
 * ========================================================================================== */
int dihip_prefill_attn(void* stream, void* out, const void* q, const void* k, const void* v,
                       int seq_q, int seq_k, int q_stride, int kv_stride, int n_heads,
                       int n_groups, int head_size, int causal, float alpha, int dtype);

/* =============================================================================================
 * 5. Glue ops for an end-to-end decode step
 * ========================================================================================== */
/* LayerNormNoBeta (csrc/core/kernel/cuda/layernorm.cu): y = (gamma*x) * rsqrt(mean(x^2)+eps)  */
int dihip_rmsnorm(void* stream, void* y, const void* x, const void* gamma, float eps, int rows,
                  int cols, int dtype);
/* Rotary on the q and k heads of fused qkv rows [rows, (n+2g)*H] in place, rotate-half
 * convention (csrc/core/kernel/cpu/rotary.cpp:22-106); positions[rows] on device; inv_freq[H/2] */
int dihip_rope_qk(void* stream, void* qkv, const uint32_t* positions, const float* inv_freq,
                  int rows, int num_heads, int num_groups, int head_size, int dtype);
/* Binary ADD / MUL and SiLU*MUL on FT tensors */
int dihip_binary_add(void* stream, void* y, const void* a, const void* b, size_t count, int dtype);
int dihip_silu_mul(void* stream, void* y, const void* gate, const void* up, size_t count, int dtype);
/* The remaining elementwise operators of the reference layer graph as stand-alone calls (the fused decode step never
 * launches them; the operator layer needs them so that an unmodified graph resolves on DeviceType::HIP):
 *   Binary MUL  (csrc/core/operator/general/binary/binary_op.cpp, BinaryType MUL = 2)
 *   Unary       (unary_op.cpp; act = allspark UnaryType, DIHIP_ACT_*)
 *   UnaryGLU    (unary_glu_op.cpp; kernel unary.cu:120-133): y[row, col] = act(x[row, col]) * x[row, inner + col], x [outer, 2 * inner]
 *   EmbeddingT5 with an FT output (embeddingT5_op.cpp): out[m, :] = table[ids[m], :], ids clamped to [0, vocab)
 *   cast FT -> f32 (greedy GenerateOp over FT logits)                                              */
int dihip_binary_mul(void* stream, void* y, const void* a, const void* b, size_t count, int dtype);
int dihip_unary(void* stream, void* y, const void* x, size_t count, int act, int dtype);
int dihip_unary_glu(void* stream, void* y, const void* x, size_t outer, size_t inner, int act, int dtype);
int dihip_embedding_ft(void* stream, void* out, const int64_t* ids, const void* table, int M, int K, int vocab,
                       int dtype);
int dihip_cast_to_f32(void* stream, float* y, const void* x, size_t count, int dtype);
/* Unquantised 16-bit weights run through the same MFMA kernel family as the weight-only GEMM:
 * the [K, N] weight is re-laid-out once into dihip tile-major order (dihip_dense_pack).
 *   dihip_gemm_a16w16 : op type "Gemm" semantics, y = act(alpha x.W + bias) (+ residual)
 *   dihip_lm_head     : logits f32 [M, N] = rmsnorm(h; gamma, eps) . W   (final LayerNormNoBeta +
 *                       lm_head Gemm of python/pyhie/allspark/model/qwen_v15.py:383-388 fused)   */
size_t dihip_dense_packed_weight_bytes(int N, int K);
int dihip_dense_pack(void* stream, const void* w_kn, int N, int K, int dtype, void* w_packed);
size_t dihip_dense_workspace_bytes(int M, int N, int K);
int dihip_gemm_a16w16(void* stream, const void* x, const void* w_packed, const void* bias,
                      const void* residual, void* y, int M, int N, int K, int act, float alpha,
                      void* ws, size_t ws_bytes, void* sync, int dtype);
int dihip_lm_head(void* stream, float* logits, const float* h, const void* gamma, float eps,
                  const void* w_packed, int M, int N, int K, void* ws, size_t ws_bytes, void* sync,
                  int dtype);
/* greedy sampling (GenerateOp top_k = 1): ids[m] = argmax_n logits[m, n] (lowest index on ties) */
int dihip_argmax(void* stream, int64_t* ids, const float* logits, int M, int N, void* ws,
                 size_t ws_bytes);
/* same, and the two per-request u32 counters (e.g. cached / total sequence length; either may be NULL)
 * advance by one in the same launch: the decode step needs no separate length-update kernels */
int dihip_argmax_advance(void* stream, int64_t* ids, const float* logits, int M, int N, void* ws,
                         size_t ws_bytes, uint32_t* counters_a, uint32_t* counters_b);
/* The sampling half of GenerateOp (generate_op.cpp:472-600; arithmetic of its x86 path, generate_impl_cpu.hpp:120-170): per row
 *   top-k (k largest logits, descending; top_k == 0 = the whole vocabulary, generate_op.cpp:338-339; rows with 1 <= top_k <= 1024 sort
 *   their candidates in LDS, rows with top_k == 0 or > 1024 run the sort-free wide kernel: see dihip_sample_rows) ->
 *   softmax(logit / T) -> top-p (shortest prefix whose cumulated probability EXCEEDS p; p <= 1e-7: off, kernel/cpu/topp.cpp) ->
 *   softmax(logit / T) over the prefix -> exponential race prob_i / -log1p(-u_i), first maximum wins (kernel/cpu/sample.cpp:42-68).
 * top_k / top_p / temperature / seed: device arrays [M].  The random stream is the backend's own, a pure function of
 * (seed[m], position[m], candidate rank -- wide rows: 0x40000000 + token index): position = device-resident index of the token being sampled (NULL: 0), so that a
 * captured step replays; counters_a / _b (may be NULL) advance by one per row like dihip_argmax_advance.  probs_out / cand_out
 * (tests, may be NULL): [M, 1024] final probabilities / candidate indices in sorted order (0 / -1 beyond the kept prefix / k). */
int dihip_sample(void* stream, int64_t* ids, const float* logits, int M, int N, const int* top_k,
                 const float* top_p, const float* temperature, const unsigned long long* seed,
                 const uint32_t* position, uint32_t* counters_a, uint32_t* counters_b, float* probs_out,
                 int* cand_out);
/* the same without the diagnostics outputs; wide_rows = how many rows have top_k == 0 (the whole vocabulary: pure top-p sampling, generate_op.cpp:
 * 338-339) or top_k > 1024 -- those rows run the sort-free wide kernel (csrc/sample.hip: the same pipeline with fixed-point masses, a radix
 * select by mass for the top-p prefix and a random stream keyed by the token index), launched only when wide_rows != 0 (-1: unknown). */
int dihip_sample_rows(void* stream, int64_t* ids, const float* logits, int M, int N, const int* top_k, const float* top_p,
                      const float* temperature, const unsigned long long* seed, const uint32_t* position, uint32_t* counters_a,
                      uint32_t* counters_b, int wide_rows);
/* The logits processors GenerateOp runs BEFORE sampling (generate_op.cpp:536-538 -> generate_impl_gpu.hpp:94-111 -> cuda::LogitsProcessor,
 * csrc/core/kernel/cuda/beam_search.cu:456-539), in place on f32 logits [M, N], per request m with the BatchGencfg lists of
 * generate_op.cpp:239-312 as device arrays [M]: repetition penalty over the distinct tokens of ids[m, lo .. cur_len) (lo = input_len when
 * suppress_repetition_in_generation, else 0; s < 0 ? s * p : s / p), frequency / presence penalty over the generated tokens
 * ids[m, input_len .. cur_len) (s -= count * frequency + (count > 0 ? presence : 0)), no-repeat n-gram and minimum length (s = -1e9).
 * ids: INT64 [M, max_len] (the reference's max_dec_ids).  ONE launch driven by the requests' tokens: O(cur_len) logits touched, no copy
 * of the scores and no memset of the count array (ws: dihip_logits_processor_workspace_bytes(M, N), scratch without state between
 * calls).  Bit-identical to LogitsProcessor<float> evaluated without contraction (oracle/logits_proc.py). */
size_t dihip_logits_processor_workspace_bytes(int M, int N);
int dihip_logits_processor(void* stream, float* logits, int M, int N, const int64_t* ids, int max_len, const int* cur_len,
                           const int* input_len, const float* repetition_penalty, const float* frequency_penalty,
                           const float* presence_penalty, const int* no_repeat_ngram_size, const int* min_length,
                           const int* eos_token_id, const int* suppress_repetition_in_generation, void* ws, size_t ws_bytes);
/* the same over DEVICE-RESIDENT histories, for a decode step that replays as a hipGraph: history_rows [M] device pointers to the requests'
 * own id buffers of max_len entries (a null row: that request asked for nothing); append_ids (may be NULL): the step's input id of every
 * request, written to history[cur_len - 1] before the processors read it -- the history grows on the device, the host sends nothing per step
 * (the reference copies every request's generated_ids_gpu into max_dec_ids each step: fill_max_dec_ids_gpu, generate_impl_gpu.hpp:270-304). */
int dihip_logits_processor_rows(void* stream, float* logits, int M, int N, int64_t* const* history_rows, const int64_t* append_ids,
                                int max_len, const int* cur_len, const int* input_len, const float* repetition_penalty,
                                const float* frequency_penalty, const float* presence_penalty, const int* no_repeat_ngram_size,
                                const int* min_length, const int* eos_token_id, const int* suppress_repetition_in_generation, void* ws,
                                size_t ws_bytes);
/* Log-probabilities AFTER sampling (generate_op.cpp:600-606 -> generate_impl_gpu.hpp:33-80 logprobs_gpu: log-softmax of the processed logits,
 * SelectBatchTokenLogprob csrc/core/kernel/cuda/logprob.cu:15-35, top-k of the log-probabilities): token_logprob[m] = log-softmax(logits[m])
 * [chosen[m]] (either may be NULL), top_value / top_index [M, out_stride]: the top_n <= 32 largest log-probabilities and their tokens
 * (value descending, lower index first on ties; -inf / -1 beyond the row's length).  Two launches that fill the chip at any batch (per-chunk
 * maxima, sums and candidates in `ws`: dihip_logprobs_workspace_bytes(M, N, top_n), 8-byte aligned, no state between calls; then one merge per
 * row); the [M, N] log-probability tensor the reference materialises is never written. */
size_t dihip_logprobs_workspace_bytes(int M, int N, int top_n);
int dihip_logprobs(void* stream, const float* logits, int M, int N, const int64_t* chosen, int top_n, int out_stride,
                   float* token_logprob, float* top_value, int* top_index, void* ws, size_t ws_bytes);
/* the same into per-request device-resident logs (graph replay: nothing returns to the host per step): row m writes the record
 * {token_logprob, top values [out_stride], top indices [out_stride] (int32)} = 1 + 2 * out_stride words at
 * records[m] + (position[m] + position_bias) * (1 + 2 * out_stride); a null records[m] or a position outside [0, max_records): skipped. */
int dihip_logprobs_records(void* stream, const float* logits, int M, int N, const int64_t* chosen, int top_n, int out_stride,
                           float* const* records, const uint32_t* position, int position_bias, int max_records, void* ws,
                           size_t ws_bytes);
/* vocabulary-parallel greedy sampling for TP: one {f32 value, i32 global index} pair per row
 * from this rank's logits slice [M, N] (global index = local + index_offset); after an
 * all-gather of the pairs ([nparts][M]) every rank merges them to the same ids.                */
int dihip_argmax_partial(void* stream, void* pairs_out, const float* logits, int M, int N,
                         int index_offset, void* ws, size_t ws_bytes);
int dihip_argmax_merge(void* stream, int64_t* ids, const void* pairs, int nparts, int M);
/* embedding lookup into the f32 hidden stream: h[m,:] = float(table[ids[m],:])                */
int dihip_embedding(void* stream, float* h, const int64_t* ids, const void* table, int M, int K,
                    int dtype);
/* the same with the ids clamped to [0, vocab): an id out of range (e.g. the arg-max of an all-NaN row upstream) reads a
 * valid row instead of faulting */
int dihip_embedding_v(void* stream, float* h, const int64_t* ids, const void* table, int M, int K, int vocab,
                      int dtype);
/* seq_lens[b] += 1 (device-side step counter for graph replay) */
int dihip_increment_u32(void* stream, uint32_t* v, int count);
/* Read-only prefetch of up to 8 buffers into the on-die Infinity Cache (no reference counterpart):
 * meant for a SIDE stream next to a latency-bound phase, so that the weight-streaming launches that
 * follow find their weights on-die.  bufs / bytes are HOST arrays of device pointers / sizes.   */
int dihip_prefetch(void* stream, const void* const* bufs_host, const size_t* bytes_host, int count,
                   int num_workgroups);

/* =============================================================================================
 * 6. Tensor-parallel all-reduce (replaces AllReduceOp's ncclAllReduce,
 *    csrc/core/operator/nccl/allreduce/allreduce_op.cpp:84-92).  `comm` is an ncclComm_t (RCCL)
 *    passed as void*.  In-place capable.  Unlike the reference it does NOT host-synchronise.
 * ========================================================================================== */
int dihip_rccl_unique_id(void* id128_host);
int dihip_rccl_comm_init_rank(void** comm, int nranks, const void* id128_host, int rank);
int dihip_rccl_comm_destroy(void* comm);
int dihip_allreduce_sum(void* comm, void* stream, const void* in, void* out, size_t count,
                        int dtype);
/* AllGatherOp (csrc/core/operator/nccl/allgather/allgather_op.cpp:40-42): every rank contributes
 * `bytes_per_rank` bytes; out holds nranks * bytes_per_rank in rank order.                     */
int dihip_allgather_bytes(void* comm, void* stream, const void* in, void* out,
                          size_t bytes_per_rank);
/* AllGatherOp (csrc/core/operator/nccl/allgather/allgather_op.cpp:27-58): every rank contributes `rows` rows of
 * row_bytes; out = row-major [rows][nranks * row_bytes] (rank r's piece of row m at column offset r * row_bytes) --
 * ncclAllGather into tmp (>= nranks * rows * row_bytes) followed by the axis-0/1 transpose.  nranks == 1: copy.  */
int dihip_allgather_rows(void* comm, void* stream, const void* in, void* tmp, void* out, int rows,
                         size_t row_bytes, int nranks);
/* its second half on its own: rank-major tmp [nranks][rows][row_bytes] -> row-major out [rows][nranks * row_bytes] */
int dihip_gather_rows_transpose(void* stream, void* out, const void* tmp, int nranks, int rows, size_t row_bytes);

/* 6b. One-shot peer-to-peer sum all-reduce for decode-sized messages (<= dihip_p2p_ar_max_bytes(): one hidden row per
 * request).  Every rank writes its row directly into a slot of every peer's receive buffer over xGMI (point-to-point
 * links: one hop, all links in parallel), raises a flag, waits for the flags addressed to itself and sums the rows in
 * rank order in f32 -- one launch and one hop instead of the 2 (n - 1) dependent hops of a ring, results bit-identical
 * on all ranks.  Same call shape as dihip_allreduce_sum (the AllReduceOp replacement,
 * csrc/core/operator/nccl/allreduce/allreduce_op.cpp:84-92); graph-capturable (the epoch lives on the device).
 * Setup, once per process group: every rank allocates a receive buffer (dihip_p2p_ar_alloc), publishes its 64-byte IPC
 * handle (dihip_ipc_get_handle; exchange by any host transport), opens the peers' handles (dihip_ipc_open_handle) and
 * builds the communicator from the nranks device pointers (its own buffer at index `rank`).  Ranks living in one
 * process (tests) pass the raw pointers.  count * sizeof(element) must be a multiple of 8; in / out 8-byte aligned.  */
size_t dihip_p2p_ar_buffer_bytes(void);
size_t dihip_p2p_ar_max_bytes(void);
int dihip_p2p_ar_alloc(void** buf);
int dihip_p2p_ar_free(void* buf);
int dihip_ipc_get_handle(void* dev_ptr, void* handle64);
int dihip_ipc_open_handle(const void* handle64, void** dev_ptr);
int dihip_ipc_close_handle(void* dev_ptr);
int dihip_p2p_ar_create(void** comm, int rank, int nranks, void* const* bufs);
int dihip_p2p_ar_destroy(void* comm);
int dihip_p2p_allreduce_sum(void* comm, void* stream, const void* in, void* out, size_t count, int dtype);
/* How long a call waits for a peer's row (polls of ~100 ns; default 2^24) and what happens then: trap != 0 (default) aborts the
 * process -- a hung collective must never hang the box; trap == 0 is for PROBING a freshly created communicator: the kernel
 * gives up, sets an error word and returns; dihip_p2p_ar_error reads it (synchronises).  After an error the communicator's
 * epochs are out of step with its peers': destroy it. */
int dihip_p2p_ar_set_timeout(void* comm, unsigned long long max_spins, int trap);
int dihip_p2p_ar_error(void* comm, int* error);

/* =============================================================================================
 * 7. Diagnostics (no reference counterpart)
 * ========================================================================================== */
/* Per-wave wall-clock stamps of the decode GEMV (gemv_stream_kernel.hpp): the next launches write
 * [workgroup][8 waves][8 stamps] uint64 (100 MHz ticks) into `buf` while it is set; NULL disables. */
int dihip_debug_set_trace(void* buf, size_t bytes);
/* launch plan of the decode GEMV for a shape (DIHIP_PARAM_ERROR when the general kernel is used) */
int dihip_debug_gemv_plan(int wbits, int M, int N, int K, int group_size, int dual, int* blocks,
                          int* upb, int* wk, int* wn, size_t* lds_bytes);
/* split plan of the decode attention for a shape on `num_cus` CUs (0: this device's count, 256 without a device) */
int dihip_debug_attn_plan(int batch, int n_heads, int n_groups, int max_seq_len, int kv_mode, int dtype, int num_cus,
                          int* nsplits, int* mfma);

#ifdef __cplusplus
}
#endif
#endif /* DASHINFER_HIP_H_ */
