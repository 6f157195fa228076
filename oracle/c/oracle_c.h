/* oracle/c/oracle_c.h -- plain-C restatement of the reference's CPU semantics.
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may link or call this library, and only as the checker (see oracle/__init__.py). */
#ifndef DASHINFER_ORACLE_C_H_
#define DASHINFER_ORACLE_C_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FT codes used across the oracle */
enum { ORC_F32 = 0, ORC_BF16 = 1, ORC_F16 = 2 };
/* KV cache quant modes (span-attention/include/spanattn/span_attn.h:36-45) */
enum { ORC_KV_NONE = 0, ORC_KV_I8 = 1, ORC_KV_U4 = 2 };

float orc_round_ft(float x, int ft);

/* ---- gemm_ref.c: tests/cpp/operator/cuda/operator_gemm_lowp_test.cpp:138-219 ------------- */
/* C[m,n] = FT(alpha * sum_k A[m,k] * ((float(B[k,n]) - Z[k/G,n]) * S[k/G,n])), sequential f32
 * accumulation over k exactly as CPU_SubC_Ref / CPU_PerC_Ref.  A,S,Z must already hold
 * FT-representable values; wbits 8: B is int8 [K,N]; wbits 4: B is packed u8 [K,ceil(N/2)].
 * group <= 0 -> per-channel.  Then (our op-level extension, gemm_a16w8_gpu.cpp:169-248):
 * + bias[n], activation = UnaryType value of csrc/proto/allspark.proto:68-76 (0 none, 1 tanh,
 * 2 gelu-erf, 3 gelu-tanh, 4 relu, 5 silu, 6 sigmoid), rounded to FT. */
int orc_gemm_a16wx(const float* A, const void* B, const float* S, const float* Z,
                   const float* bias, float* C, int M, int N, int K, int group, int wbits,
                   float alpha, int act, int ft);

/* "x86 medium_bf16" semantics (csrc/core/operator/general/gemm/gemm_op_cpu.cpp:75-126):
 * bf16(x) . bf16((q-z)*s), f32 accumulate, f32 out (no FT rounding of the result). */
int orc_gemm_a16wx_x86bf16(const float* A, const void* B, const float* S, const float* Z,
                           const float* bias, float* C, int M, int N, int K, int group,
                           int wbits, float alpha, int act);

/* ---- kv_codec.c: span-attention/src/cache_quant/impl_i8.cuh:53-66,116-142,
 *                  impl_u4.cuh:79-103,157-184; layout decoder_cache_append.cuh:33-87 -------- */
size_t orc_span_bytes(int g, int S, int H, int mode, int ft);
/* write one token-head (H values) at position pos of a span: data [g][S][H'] then params
 * [g][S]{f32 zero, f32 scale} (quantised modes only) */
void orc_span_write_head(void* span, const float* x, int head, int pos, int g, int S, int H,
                         int mode, int ft);
/* read it back dequantised to f32 */
void orc_span_read_head(const void* span, float* x, int head, int pos, int g, int S, int H,
                        int mode, int ft);

/* ---- attention.c: csrc/core/operator/generate_opt/batch_mqa/batch_mqa_op.cpp:140-179 ------ */
/* decode attention for one request over paged spans: q [n,H] f32, out [n,H] f32 (not rounded) */
void orc_span_attn_decode(float* out, const float* q, const void* const* kspans,
                          const void* const* vspans, int len, int n, int g, int H, int S, int mode,
                          int ft, float alpha);
/* causal prefill, GQA: q [Lq, n, H], k/v [Lk, g, H] f32; query i attends keys <= i + (Lk-Lq) */
void orc_prefill_attn(float* out, const float* q, const float* k, const float* v, int Lq, int Lk,
                      int n, int g, int H, float alpha, int causal);

#ifdef __cplusplus
}
#endif
#endif
