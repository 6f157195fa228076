// span_attn_common.hpp -- device helpers shared by the decode attention kernels
// (span_attn.hip: op-boundary form; span_attn_fused.hip: Rotary + cache append + attention).
#pragma once
#include "device_utils.h"

namespace dihip {

constexpr int ATTN_THREADS = 256;
constexpr int ATTN_TB = 2;              // tokens per lane slot per iteration
constexpr int ATTN_TOK_PER_ITER = 32;  // 4 waves x 4 token slots x ATTN_TB tokens
constexpr int ATTN_PSTRIDE = 132;      // floats per partial record: o[128], m, l, pad

struct AttnArgs {
  void* out;
  const void* q;
  const void* const* kspans;
  const void* const* vspans;
  const uint32_t* seq_lens;  // including the new token
  float* partials;           // [B*n][nsplits][ATTN_PSTRIDE]
  unsigned* counters;        // [B][g*nchunks]
  int B, n, g, hpg, S, span_stride, nsplits, nchunks;
  float scale;
  int out_frag_mt;  // 0: out is row-major [B, n*H]; 1 / 2: FRAG32 with that many 16-row tiles (act_frag_index)
  // decode-step form (Rotary + DecoderCacheAppend folded in): q points at the fused pre-Rotary qkv rows
  // [B, (n + 2g) * H], seq_lens holds the tokens ALREADY cached (position of the new token)
  const float* rope_tab;  // [max_pos][64] {cos, sin}; non-null selects the decode-step form
  int len_bias;           // sequence length = seq_lens[b] + len_bias (decode step on the op-boundary kernels: lengths BEFORE the append, + 1)
  int force_partials;     // write the block's partial record even for a single split and leave the merge to the caller
  int tps_static;         // decode-step form: tokens per split fixed by the host from max_seq_len (0: from the request's
                          // length, as the op-boundary kernels).  A split's token range -- and with it the span-table
                          // entries of a wave's first tiles -- then depends on the kernel arguments alone: the pointer loads
                          // go out together with the length load instead of one round trip behind it.
  int merge_wt;           // in-launch merge of the split partials (a.counters != null): records stored write-through,
                          // arrival ticket, the last workgroup of a (request, group) reads all records past its L1 and
                          // writes the output -- no release / acquire fences, no second launch
  size_t partial_bytes;   // size of `partials` (buffer-resource range of the write-through stores / loads)
  unsigned long long* trace;  // diagnostics (`make trace` build + dihip_debug_set_trace): [workgroup][wave][8] wall-clock stamps, or null
};

// In-launch hand-off of the fused attention block (decode_attn_block.hip: RMSNorm + qkv GEMV, Rotary + append + attention and
// the o-projection in ONE launch).  Transport: 8-byte granules {value (low dword), tag (high dword)} written by ONE aligned
// agent-scope store and polled with agent-scope loads -- the data is the flag (MI355X guide, Guideline 16 form R2): a stale
// line can only show an older tag.  `tag` is the launch's epoch (read from the state block at entry, never 0).
struct AttnHandoff {
  const unsigned long long* qkv_gran;  // [(n + 2g) * H]: one FT element of the fused pre-Rotary qkv row per granule
  unsigned long long* out_gran;        // [n * H / 2]: two packed FT elements of the attention output per granule
  size_t out_gran_bytes;
  unsigned* grp_flag;                  // [g]: tag once the group's merged output granules have drained
  unsigned* err;                       // set non-zero when a bounded wait gave up (results are then garbage, nothing hangs)
  unsigned tag;
  unsigned spin_limit;
  // polled split records (round 6; null: write-through records + arrival ticket in AttnArgs::partials): TWO buffers of
  // [n][nsplits][ATTN_PSTRIDE] words (rec_bytes each) inside the block's own sync buffer, used alternately by launch parity, zero
  // before use.  See merge_polled_items (span_attn_ft_mfma.hpp).
  unsigned* rec;
  unsigned rec_bytes;
  unsigned parity;
};

// decode-step form on the matrix cores (span_attn.hip); returns a DIHIP status, DIHIP_PARAM_ERROR with
// *handled = false when the configuration is not covered (caller falls back to its own kernels)
int span_attn_fused_mfma(void* stream, void* output, const void* qkv, void* const* k_span_array, void* const* v_span_array,
                         const uint32_t* old_seq_lens_dev, const float* rope_table, int batch, int n_heads, int n_groups,
                         int span_len, int n_spans_per_request, int max_seq_len, int kv_mode, int dtype, float qk_scale, void* ws,
                         size_t ws_bytes, bool* handled, void* sync = nullptr, size_t sync_bytes = 0, int out_layout = 0);
size_t span_attn_fused_mfma_workspace_bytes(int batch, int n_heads, int n_groups, int max_seq_len);
// the op-boundary decode kernels (span_attn.hip: run_decode) on lengths seq_lens[b] + len_bias; returns a DIHIP status
int span_attn_decode_biased(void* stream, void* output, const void* query, const void* const* k_span_array,
                            const void* const* v_span_array, const uint32_t* seq_lens_dev, int len_bias, int batch, int n_heads,
                            int n_groups, int span_len, int n_spans_per_request, int max_seq_len, int kv_mode, int dtype,
                            float qk_scale, void* ws, size_t ws_bytes);
size_t span_attn_decode_workspace_bytes(int batch, int n_heads, int n_groups, int max_seq_len, int kv_mode, int dtype);
// Rotary (cos / sin from the dihip_rope_table table) + cache append + rotated q, as dihip_rope_kv_append (span_cache.hip)
int rope_table_kv_append(void* stream, void* const* k_spans, void* const* v_spans, void* q_out, const void* qkv,
                         const uint32_t* old_seq_lens, const float* rope_table, int batch, int num_heads, int num_groups,
                         int span_len, int span_stride, int kv_mode, int dtype);

// sum over the 16 lanes of a DPP row (all 16 lanes receive the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, true));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true));  // row_ror:1
  return v;
}

template <typename T, bool GL>
__device__ __forceinline__ T kv_ld(const void* p) {
  if constexpr (GL) return gload<T>(p);
  else return *reinterpret_cast<const T*>(p);
}

// 8 consecutive head dims (d = dc*8 ..) of one token-head, dequantised to f32
template <int FT, int MODE>
struct KvChunk {
  u32x4_t raw0, raw1;  // raw1 only for f32
  float zero, scale;
};

// GL: load through the global address space (exact vmcnt accounting, see gload) or as plain (FLAT) loads
template <int FT, int MODE, bool GL = true>
__device__ __forceinline__ void kv_issue(KvChunk<FT, MODE>& c, const void* span, int grp, int pos, int g, int S, int dc) {
  constexpr int H = 128;
  if constexpr (MODE == DIHIP_KV_NONE) {
    if constexpr (FT == DIHIP_F32) {
      const float* p = reinterpret_cast<const float*>(span) + ((size_t)grp * S + pos) * H + dc * 8;
      c.raw0 = kv_ld<u32x4_t, GL>(p);
      c.raw1 = kv_ld<u32x4_t, GL>(p + 4);
    } else {
      c.raw0 = kv_ld<u32x4_t, GL>(reinterpret_cast<const uint16_t*>(span) + ((size_t)grp * S + pos) * H + dc * 8);
    }
  } else {
    constexpr int HB = MODE == DIHIP_KV_I8 ? H : H / 2;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(span);
    const unsigned char* d = base + ((size_t)grp * S + pos) * HB;
    if constexpr (MODE == DIHIP_KV_I8) {
      const u32x2_t v = kv_ld<u32x2_t, GL>(d + dc * 8);
      c.raw0 = u32x4_t{v[0], v[1], 0, 0};
    } else {
      c.raw0 = u32x4_t{kv_ld<uint32_t, GL>(d + dc * 4), 0, 0, 0};
    }
    const float* params = reinterpret_cast<const float*>(base + (size_t)g * S * HB) + ((size_t)grp * S + pos) * 2;
    const u32x2_t pz = kv_ld<u32x2_t, GL>(params);
    c.zero = __uint_as_float(pz[0]);
    c.scale = __uint_as_float(pz[1]);
  }
}

template <int FT, int MODE>
__device__ __forceinline__ void kv_decode(const KvChunk<FT, MODE>& c, float (&x)[8]) {
  if constexpr (MODE == DIHIP_KV_NONE) {
    if constexpr (FT == DIHIP_F32) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = __uint_as_float(c.raw0[j]);
        x[4 + j] = __uint_as_float(c.raw1[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[2 * j] = ft_bits_to_f32<FT == DIHIP_F32 ? DIHIP_BF16 : FT>(c.raw0[j] & 0xFFFFu);
        x[2 * j + 1] = ft_bits_to_f32<FT == DIHIP_F32 ? DIHIP_BF16 : FT>(c.raw0[j] >> 16);
      }
    }
  } else if constexpr (MODE == DIHIP_KV_I8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = (int)(signed char)((c.raw0[j >> 2] >> (8 * (j & 3))) & 0xFFu);
      x[j] = ((float)q - c.zero) * c.scale;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned q = (c.raw0[0] >> (4 * j)) & 0xFu;  // lo nibble = even d (impl_u4.cuh:27-36)
      x[j] = ((float)q - c.zero) * c.scale;
    }
  }
}

__device__ __forceinline__ float safe_exp_diff(float a, float b) {  // exp(a - b), 0 when a == -inf
  return a == -INFINITY ? 0.f : __expf(a - b);
}

// merge_order4 -- THE order in which the partial records (o[128], m, l) of a split sequence are merged, in every implementation
// (span_attn_split_merge_kernel, merge_split_records, merge_polled_items4: bit-identical outputs):
//   M   = max_j m_j                                      over all splits
//   c_j = safe_exp_diff(m_j, M)
//   s_r = fma chain over j = r, r + 4, r + 8, ...  (ascending), from 0:  s_r = fmaf(x_j, c_j, s_r)        r = 0 .. 3,  x = l or an element of o
//   sum = (s_0 + s_1) + (s_2 + s_3);   out = l_sum > 0 ? o_sum / l_sum : 0
// Four chains so that a quad of lanes can share the splits of one item -- each lane polls and multiplies a quarter of the records of the
// fused attention block's hand-off, two DPP steps combine -- and so that a sequential merge has four independent chains.


}  // namespace dihip
