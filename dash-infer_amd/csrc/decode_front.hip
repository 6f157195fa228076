// decode_front.hip -- the front half of a decode layer in ONE launch: RMSNorm + qkv GEMV (+bias) and Rotary + cache append +
// paged attention (include/dashinfer_hip.h section 3d).
//
// In the launch chain the decode attention is pure latency: four dependent round trips (sequence length, span pointer,
// K / V rows, then the short compute) on 68 of 256 CUs behind a kernel boundary, 7.2 us + 1.3 us of boundary for 4 MB.
// Three of those round trips do not depend on q.  Here the attention workgroups are part of the qkv GEMV's launch: they
// resolve lengths and span pointers and pull their K / V tiles while the GEMV workgroups stream the qkv weights, then
// wait on a per-(request, KV group) counter that the GEMV workgroups bump as they publish their column tiles (16-bit
// stores written through, agent scope), read this step's q / k / v with agent-scope loads and finish: what is left
// behind the GEMV is the attention's compute tail.  Replaces, per layer, the Gemm[A16W8|A16W4](qkv) + Rotary + DecOptMQA
// sequence of the reference graph (qwen_v15.py:218-262) -- three operators, five to six launches there.
//
// Co-residency: the GEMV workgroups have the low block indices (dispatched first) and never wait; the attention workgroups
// (at most one per CU next to a GEMV workgroup: 8 + 4 waves, 2 x 96 + 256 VGPRs per SIMD lane budget, < 64 KB of LDS
// together) only wait.  A wait that sees no progress for tens of seconds traps.
#include <algorithm>

#include "gemv_stream_kernel.hpp"
#include "span_attn_ft_mfma.hpp"

namespace dihip {

bool gemv_front_plan(int wbits, const float* h, const void* gamma, float eps, const void* w_packed, const void* sz_packed,
                     const void* bias, void* y, int M, int N, int K, int group_size, GemvArgs* g_out, int* blocks, size_t* lds_bytes,
                     int* mr, int* gpt);
int span_attn_front_plan(int batch, int n_heads, int n_groups, int max_seq_len, int kv_mode, int dtype, int* nsplits, int* nchunks,
                         size_t* partial_bytes);

template <int WBITS, int MR, int GPT>
__global__ __launch_bounds__(GEMV_THREADS, 2) void decode_front_kernel(const GemvArgs g, const AttnArgs a, const int gemv_blocks,
                                                                    const int attn_gx, const int attn_gy, const int attn_gz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bid = (int)blockIdx.x;
  if (bid < gemv_blocks) {
    gemv_stream_body<WBITS, DIHIP_BF16, MR, PRO_RMSNORM, EPI_STD, GPT, false, GEMV_SYNC_PUB>(g, bid, gemv_blocks, smem);
  } else {
    if (threadIdx.x >= ATTN_THREADS) return;  // the attention body is a 4-wave workgroup (barriers count live waves only)
    const int i = bid - gemv_blocks;
    span_attn_ft_mfma_body<DIHIP_BF16, DIHIP_KV_NONE, true, true>(a, i % attn_gx, (i / attn_gx) % attn_gy, i / (attn_gx * attn_gy),
                                                                   attn_gx, attn_gy, attn_gz, smem);
  }
}

template <int WBITS, int MR, int GPT>
static hipError_t launch_front(const GemvArgs& g, const AttnArgs& a, int gemv_blocks, int gx, int gy, int gz, size_t lds, hipStream_t s) {
  auto kern = decode_front_kernel<WBITS, MR, GPT>;
  hipLaunchKernelGGL(kern, dim3(gemv_blocks + gx * gy * gz), dim3(GEMV_THREADS), lds, s, g, a, gemv_blocks, gx, gy, gz);
  return hipGetLastError();
}

static bool front_enabled() {
  static int enabled = -1;  // DIHIP_DECODE_FRONT=0: report "not supported" (callers keep the launch chain)
  if (enabled < 0) {
    const char* e = getenv("DIHIP_DECODE_FRONT");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  return enabled != 0;
}

}  // namespace dihip

using namespace dihip;

extern "C" {

size_t dihip_decode_front_sync_bytes(int batch, int n_groups) {
  return batch > 0 && n_groups > 0 ? (size_t)2 * batch * n_groups * FRONT_SYNC_STRIDE * sizeof(unsigned) : 0;
}

int dihip_decode_front_supported(int wbits, int M, int K, int group_size, int n_heads, int n_groups, int head_size, int max_seq_len,
                                 int kv_mode, int dtype) {
  if (!front_enabled() || head_size != 128 || M < 1 || M > 4 || n_heads <= 0 || n_groups <= 0 || n_heads % n_groups) return 0;
  int ns, nc;
  size_t pb;
  if (!span_attn_front_plan(M, n_heads, n_groups, max_seq_len, kv_mode, dtype, &ns, &nc, &pb)) return 0;
  GemvArgs g;
  int blocks, mr, gpt;
  size_t lds;
  alignas(16) static const float dummy[4] = {0};
  const int N = (n_heads + 2 * n_groups) * 128;
  if (!gemv_front_plan(wbits, dummy, dummy, 1e-6f, dummy, dummy, nullptr, (void*)dummy, M, N, K, group_size, &g, &blocks, &lds, &mr, &gpt)) return 0;
  int ncu = cached_num_cus();
  if (ncu <= 0) ncu = 256;
  // every workgroup of the launch must be resident at once: GEMV workgroups one per CU, attention workgroups at most two per CU
  return blocks <= ncu && (long)ns * n_groups * nc * M <= 2L * ncu ? 1 : 0;
}

size_t dihip_decode_front_workspace_bytes(int batch, int n_heads, int n_groups, int max_seq_len) {
  int ns, nc;
  size_t pb = 0;
  if (batch <= 0 || n_heads <= 0 || n_groups <= 0 || n_heads % n_groups) return 0;
  if (!span_attn_front_plan(batch, n_heads, n_groups, max_seq_len, DIHIP_KV_NONE, DIHIP_BF16, &ns, &nc, &pb)) return 0;
  return pb + 256;
}

int dihip_decode_front(void* stream, int wbits, const float* h, const void* gamma, float eps, const void* w_packed,
                       const void* sz_packed, const void* bias, void* qkv, void* attn_out, int M, int K, int group_size,
                       void* const* k_span_array, void* const* v_span_array, const uint32_t* old_seq_lens_dev,
                       const float* rope_table, int n_heads, int n_groups, int head_size, int span_len, int n_spans_per_request,
                       int max_seq_len, int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes, void* sync,
                       size_t sync_bytes) {
  DIHIP_REQUIRE(h && gamma && w_packed && sz_packed && qkv && attn_out && k_span_array && v_span_array && old_seq_lens_dev && rope_table &&
                    ws && sync,
                DIHIP_PARAM_ERROR, "decode_front: null pointer");
  DIHIP_REQUIRE(span_len == 16 || span_len == 32 || span_len == 64 || span_len == 128, DIHIP_PARAM_ERROR,
                "span_attn: span length %d not in {16,32,64,128}", span_len);
  DIHIP_REQUIRE(n_spans_per_request > 0 && max_seq_len > 0, DIHIP_PARAM_ERROR, "decode_front: invalid parameter");
  DIHIP_REQUIRE(dihip_decode_front_supported(wbits, M, K, group_size, n_heads, n_groups, head_size, max_seq_len, kv_mode, dtype), DIHIP_PARAM_ERROR,
                "decode_front: configuration not covered (M <= 4, bf16, 16-bit cache, head size 128, decode-GEMV shapes); see _supported");
  DIHIP_REQUIRE(sync_bytes >= dihip_decode_front_sync_bytes(M, n_groups), DIHIP_MEMORY_ERROR, "decode_front: sync buffer too small");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int N = (n_heads + 2 * n_groups) * 128;
  GemvArgs g;
  int blocks, mr, gpt;
  size_t lds;
  DIHIP_REQUIRE(gemv_front_plan(wbits, h, gamma, eps, w_packed, sz_packed, bias, qkv, M, N, K, group_size, &g, &blocks, &lds, &mr, &gpt),
                DIHIP_PARAM_ERROR, "decode_front: the qkv GEMV is not served by the decode GEMV (alignment)");
  int nsplits, nchunks;
  size_t pbytes;
  span_attn_front_plan(M, n_heads, n_groups, max_seq_len, kv_mode, dtype, &nsplits, &nchunks, &pbytes);
  DIHIP_REQUIRE(ws_bytes >= pbytes, DIHIP_MEMORY_ERROR, "decode_front: workspace too small (%zu < %zu)", ws_bytes, pbytes);
  unsigned* words = reinterpret_cast<unsigned*>(sync);
  g.front_counter = words;
  g.front_n = n_heads;
  g.front_g = n_groups;
  g.front_hpg = n_heads / n_groups;
  AttnArgs a{};
  a.out = attn_out;
  a.q = qkv;
  a.kspans = k_span_array;
  a.vspans = v_span_array;
  a.seq_lens = old_seq_lens_dev;
  a.partials = reinterpret_cast<float*>(ws);
  a.B = M;
  a.n = n_heads;
  a.g = n_groups;
  a.hpg = n_heads / n_groups;
  a.S = span_len;
  a.span_stride = n_spans_per_request;
  a.nsplits = nsplits;
  a.nchunks = nchunks;
  a.scale = qk_scale;
  a.rope_tab = rope_table;
  a.force_partials = 1;
  a.front_counter = words;
  a.front_done = words + (size_t)M * n_groups * FRONT_SYNC_STRIDE;
  {
    static int presleep = -1;  // DIHIP_FRONT_PRESLEEP: s_sleep(127) repetitions (~3.9 us each) before the first poll
    if (presleep < 0) {
      const char* e = getenv("DIHIP_FRONT_PRESLEEP");
      presleep = e ? std::max(0, atoi(e)) : 1;
    }
    a.front_presleep = presleep;
  }
  a.front_target = (unsigned)((a.hpg + 2) * 8);  // 8 column tiles of 16 per 128-wide head: hpg query heads + K + V
  int gx = nsplits, gy = n_groups * nchunks, gz = M;
  {
    // timing diagnostics only (results are wrong with either): DIHIP_FRONT_DEBUG=noattn launches the GEMV workgroups alone,
    // =nowait lets the attention workgroups run without waiting for the qkv row
    static int dbg = -1;
    if (dbg < 0) {
      const char* e = getenv("DIHIP_FRONT_DEBUG");
      dbg = !e ? 0 : (e[0] == 'n' && e[2] == 'a') ? 1 : 2;
    }
    if (dbg == 1) gz = 0;
    if (dbg == 2) a.front_target = 0;
  }
  const size_t lds_all = std::max<size_t>(lds, FT_MFMA_SMEM_BYTES);
  hipError_t e = hipErrorInvalidValue;
#define FRONT_GO(W_, MR_, G_) \
  if (wbits == W_ && mr == MR_ && gpt == G_) e = launch_front<W_, MR_, G_>(g, a, blocks, gx, gy, gz, lds_all, s);
  FRONT_GO(4, 1, 1) FRONT_GO(4, 1, 0) FRONT_GO(4, 4, 1) FRONT_GO(4, 4, 0) FRONT_GO(8, 1, 1) FRONT_GO(8, 1, 0) FRONT_GO(8, 4, 1) FRONT_GO(8, 4, 0)
#undef FRONT_GO
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "decode_front: launch failed: %s", hipGetErrorString(e));
  return dihip_span_attn_merge_partials(stream, attn_out, a.partials, M, n_heads, nsplits, dtype);
}

}  // extern "C"
