// span_cache.hip -- KV span writers/readers for gfx950 (include/dashinfer_hip.h section 2).
//
// Replaces csrc/core/kernel/cuda/cache/{decoder_cache_append,context_span_copy,
// prefix_cache_copy,mass_span_copy}.cuh.  The span byte layout and the quantiser arithmetic are
// bit-compatible with the reference so that its host-side cache managers (frame / span /
// prefix-cache) keep working unchanged:
//   data  [g][S][H*bits/8]      (u4: lo nibble = even d, impl_u4.cuh:27-29)
//   param [g][S]{f32 zero, f32 scale} after the data (quantised modes), one pair per token-head
//   scale = max((max-min)/RANGE, 1e-5); zero = rint(clamp(ORIGIN - min/scale)); q = rint(clamp(
//   zero + x/scale))   (impl_i8.cuh:53-66,116-142, impl_u4.cuh:79-103,157-184; IEEE division
//   instead of __fdividef).
// One 64-lane wavefront owns one token-head (the reference uses one 32-lane warp): each lane
// holds H/64 consecutive elements, min/max are wave reductions.
#include <algorithm>
#include <type_traits>

#include "device_utils.h"
#include "span_codec.hpp"

namespace dihip {

// ---- DecoderCacheAppend (decoder_cache_append.cuh:100-154): grid (batch, n), one wave ---------
template <int FT, int MODE, int EPL>
__global__ __launch_bounds__(64) void kv_append_kernel(void* const* k_spans, void* const* v_spans, void* q_out,
                                                       const void* qkv, const uint32_t* old_seq_lens, int n, int g,
                                                       int S, int span_stride) {
  constexpr int H = 64 * EPL;
  const int b = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  const size_t row = (size_t)b * (n + 2 * g) * H;
  // query pass-through
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    const float v = load_ft<FT>(qkv, row + (size_t)head * H + lane * EPL + i);
    store_ft<FT>(q_out, ((size_t)b * n + head) * H + lane * EPL + i, v);
  }
  if (head >= g) return;
  const uint32_t old_len = old_seq_lens[b];
  const uint32_t span_idx = old_len / S, pos = old_len % S;
  if (span_idx >= (uint32_t)span_stride) return;  // past the request's span table (a replay beyond max_len): drop the write
  float kx[EPL], vx[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    kx[i] = load_ft<FT>(qkv, row + (size_t)(n + head) * H + lane * EPL + i);
    vx[i] = load_ft<FT>(qkv, row + (size_t)(n + g + head) * H + lane * EPL + i);
  }
  store_token_head<FT, MODE, EPL>(k_spans[(size_t)b * span_stride + span_idx], kx, head, pos, g, S, H, lane);
  store_token_head<FT, MODE, EPL>(v_spans[(size_t)b * span_stride + span_idx], vx, head, pos, g, S, H, lane);
}

// ---- Rotary + DecoderCacheAppend fused (decode step glue; H == 128, EPL == 2) --------------------
// Same outputs as dihip_rope_qk followed by dihip_kv_append: q and k heads are rotated
// (rotate-half, csrc/core/kernel/cpu/rotary.cpp:22-106) at position old_seq_lens[b]; the rotated
// values are rounded to FT once (as the Rotary op's FT output tensor would be) before the cache
// quantiser sees them.
template <int FT, int MODE>
__global__ __launch_bounds__(64) void rope_kv_append_kernel(void* const* k_spans, void* const* v_spans, void* q_out,
                                                            const void* qkv, const uint32_t* old_seq_lens,
                                                            const float* __restrict__ inv_freq, int n, int g, int S,
                                                            int span_stride, const float* __restrict__ rope_tab = nullptr) {
  constexpr int EPL = 2, H = 128;
  const int b = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  const size_t row = (size_t)b * (n + 2 * g) * H;
  const uint32_t old_len = old_seq_lens[b];
  // lane holds d = 2*lane, 2*lane+1; the rotate-half partner (d +- 64) lives in lane ^ 32
  float cs[EPL], sn[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    // rope_tab: the same values, precomputed per position by dihip_rope_table ({cos, sin} pairs, 64 per position)
    if (rope_tab) {
      const float2 t = reinterpret_cast<const float2*>(rope_tab)[(size_t)old_len * 64 + ((lane * EPL + i) & 63)];
      cs[i] = t.x;
      sn[i] = t.y;
    } else {
      rope_sincos(old_len, inv_freq[(lane * EPL + i) & 63], &sn[i], &cs[i]);
    }
  }
  auto rotate = [&](float (&x)[EPL]) {
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const float partner = __shfl_xor(x[i], 32, 64);
      const float r = lane < 32 ? x[i] * cs[i] - partner * sn[i] : x[i] * cs[i] + partner * sn[i];
      x[i] = FT == DIHIP_F32 ? r : ft_round<FT == DIHIP_F32 ? DIHIP_BF16 : FT>(r);
    }
  };
  float qx[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) qx[i] = load_ft<FT>(qkv, row + (size_t)head * H + lane * EPL + i);
  rotate(qx);
#pragma unroll
  for (int i = 0; i < EPL; ++i) store_ft<FT>(q_out, ((size_t)b * n + head) * H + lane * EPL + i, qx[i]);
  if (head >= g) return;
  const uint32_t span_idx = old_len / S, p = old_len % S;
  if (span_idx >= (uint32_t)span_stride) return;  // past the request's span table: drop the write
  float kx[EPL], vx[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    kx[i] = load_ft<FT>(qkv, row + (size_t)(n + head) * H + lane * EPL + i);
    vx[i] = load_ft<FT>(qkv, row + (size_t)(n + g + head) * H + lane * EPL + i);
  }
  rotate(kx);
  store_token_head<FT, MODE, EPL>(k_spans[(size_t)b * span_stride + span_idx], kx, head, p, g, S, H, lane);
  store_token_head<FT, MODE, EPL>(v_spans[(size_t)b * span_stride + span_idx], vx, head, p, g, S, H, lane);
}

// ---- ContextSpanCopy (context_span_copy.cuh:49-108): grid (seq_len, g), one wave ---------------
template <int FT, int MODE, int EPL>
__global__ __launch_bounds__(64) void kv_context_copy_kernel(void* const* spans, const void* src, int src_stride,
                                                             int start_pos, int g, int S) {
  constexpr int H = 64 * EPL;
  const int t = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  float x[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) x[i] = load_ft<FT>(src, (size_t)t * src_stride + (size_t)head * H + lane * EPL + i);
  const int tok = start_pos + t;
  store_token_head<FT, MODE, EPL>(spans[tok / S], x, head, tok % S, g, S, H, lane);
}

// ---- PrefixCacheCopy (prefix_cache_copy.cuh:37-106): spans -> contiguous, dequantised ----------
template <int FT, int MODE, int EPL>
__global__ __launch_bounds__(64) void kv_prefix_gather_kernel(void* dst, void* const* spans, int g, int S) {
  constexpr int H = 64 * EPL;
  const int t = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  float x[EPL];
  load_token_head<FT, MODE, EPL>(spans[t / S], x, head, t % S, g, S, H, lane);
#pragma unroll
  for (int i = 0; i < EPL; ++i) store_ft<FT>(dst, ((size_t)t * g + head) * H + lane * EPL + i, x[i]);
}

// ---- mass span copy (mass_span_copy.cuh:30-100): raw bytes ------------------------------------
template <bool GATHER>
__global__ __launch_bounds__(256) void span_mass_copy_kernel(unsigned char* cont, void* const* spans, size_t span_bytes) {
  const int sp = blockIdx.y;
  unsigned char* span = reinterpret_cast<unsigned char*>(spans[sp]);
  unsigned char* c = cont + (size_t)sp * span_bytes;
  const size_t nvec = span_bytes / 16;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    if (GATHER) reinterpret_cast<u32x4_t*>(c)[i] = reinterpret_cast<const u32x4_t*>(span)[i];
    else reinterpret_cast<u32x4_t*>(span)[i] = reinterpret_cast<const u32x4_t*>(c)[i];
  }
  for (size_t i = nvec * 16 + (size_t)blockIdx.x * 256 + threadIdx.x; i < span_bytes; i += (size_t)gridDim.x * 256) {
    if (GATHER) c[i] = span[i];
    else span[i] = c[i];
  }
}

// dispatch over (FT, MODE, H): H == 128 for every mode; H == 64 unquantised only
template <int V>
using IC = std::integral_constant<int, V>;

template <typename F>
static bool kv_dispatch(int ft, int mode, int H, F&& f) {
  bool done = false;
  auto go = [&](auto ftc, auto modec, auto eplc) {
    if (!done && ft == decltype(ftc)::value && mode == decltype(modec)::value && H == 64 * decltype(eplc)::value) {
      f(ftc, modec, eplc);
      done = true;
    }
  };
  go(IC<DIHIP_BF16>{}, IC<DIHIP_KV_NONE>{}, IC<2>{});
  go(IC<DIHIP_BF16>{}, IC<DIHIP_KV_I8>{}, IC<2>{});
  go(IC<DIHIP_BF16>{}, IC<DIHIP_KV_U4>{}, IC<2>{});
  go(IC<DIHIP_F16>{}, IC<DIHIP_KV_NONE>{}, IC<2>{});
  go(IC<DIHIP_F16>{}, IC<DIHIP_KV_I8>{}, IC<2>{});
  go(IC<DIHIP_F16>{}, IC<DIHIP_KV_U4>{}, IC<2>{});
  go(IC<DIHIP_F32>{}, IC<DIHIP_KV_NONE>{}, IC<2>{});
  go(IC<DIHIP_F32>{}, IC<DIHIP_KV_I8>{}, IC<2>{});
  go(IC<DIHIP_F32>{}, IC<DIHIP_KV_U4>{}, IC<2>{});
  go(IC<DIHIP_BF16>{}, IC<DIHIP_KV_NONE>{}, IC<1>{});
  go(IC<DIHIP_F16>{}, IC<DIHIP_KV_NONE>{}, IC<1>{});
  go(IC<DIHIP_F32>{}, IC<DIHIP_KV_NONE>{}, IC<1>{});
  if (!done) set_last_error("kv cache: unsupported (dtype=%d, mode=%d, head_size=%d)", ft, mode, H);
  return done;
}

static bool span_len_ok(int S) { return S == 16 || S == 32 || S == 64 || S == 128; }

}  // namespace dihip

using namespace dihip;

extern "C" {

size_t dihip_span_bytes(int num_groups, int span_len, int head_size, int kv_mode, int dtype) {
  // CacheUtils::GetSpanSizeInBytes, csrc/runtime/cache/virtual_cache.cpp:202-232
  if (num_groups <= 0 || span_len <= 0 || head_size <= 0) return 0;
  size_t data = span_data_bytes(num_groups, span_len, head_size, kv_mode, dtype);
  size_t extra = kv_mode == DIHIP_KV_NONE ? 0 : (size_t)2 * span_len * num_groups * sizeof(float);
  return data + extra;
}

int dihip_kv_append(void* stream, void* const* k_spans, void* const* v_spans, void* q_out, const void* qkv,
                    const uint32_t* old_seq_lens, int batch, int num_heads, int num_groups, int head_size,
                    int span_len, int span_stride, int kv_mode, int dtype) {
  DIHIP_REQUIRE(batch >= 0 && num_heads > 0 && num_groups > 0 && span_stride > 0, DIHIP_PARAM_ERROR,
                "kv_append: bad shape");
  DIHIP_REQUIRE(num_groups <= num_heads, DIHIP_PARAM_ERROR,
                "DecoderCacheAppend: nGroups should be no more than nHeads");  // decoder_cache_append.cuh:172-175
  DIHIP_REQUIRE(span_len_ok(span_len), DIHIP_PARAM_ERROR, "kv_append: span size %d not in {16,32,64,128}", span_len);
  DIHIP_REQUIRE(k_spans && v_spans && q_out && qkv && old_seq_lens, DIHIP_PARAM_ERROR, "kv_append: null pointer");
  if (batch == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool ok = kv_dispatch(dtype, kv_mode, head_size, [&](auto ftc, auto mc, auto ec) {
    constexpr int FT = decltype(ftc)::value, MODE = decltype(mc)::value, EPL = decltype(ec)::value;
    hipLaunchKernelGGL((kv_append_kernel<FT, MODE, EPL>), dim3(batch, num_heads), dim3(64), 0, s, k_spans, v_spans,
                       q_out, qkv, old_seq_lens, num_heads, num_groups, span_len, span_stride);
  });
  if (!ok) return DIHIP_PARAM_ERROR;
  return launch_status();
}

int dihip_rope_kv_append(void* stream, void* const* k_spans, void* const* v_spans, void* q_out, const void* qkv,
                         const uint32_t* old_seq_lens, const float* inv_freq, int batch, int num_heads, int num_groups,
                         int head_size, int span_len, int span_stride, int kv_mode, int dtype) {
  DIHIP_REQUIRE(batch >= 0 && num_heads > 0 && num_groups > 0 && span_stride > 0, DIHIP_PARAM_ERROR,
                "rope_kv_append: bad shape");
  DIHIP_REQUIRE(num_groups <= num_heads, DIHIP_PARAM_ERROR, "rope_kv_append: nGroups should be no more than nHeads");
  DIHIP_REQUIRE(head_size == 128, DIHIP_PARAM_ERROR, "rope_kv_append: head size %d (only 128)", head_size);
  DIHIP_REQUIRE(span_len_ok(span_len), DIHIP_PARAM_ERROR, "rope_kv_append: span size %d", span_len);
  DIHIP_REQUIRE(k_spans && v_spans && q_out && qkv && old_seq_lens && inv_freq, DIHIP_PARAM_ERROR,
                "rope_kv_append: null pointer");
  if (batch == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool ok = kv_dispatch(dtype, kv_mode, head_size, [&](auto ftc, auto mc, auto ec) {
    constexpr int FT = decltype(ftc)::value, MODE = decltype(mc)::value;
    (void)ec;
    hipLaunchKernelGGL((rope_kv_append_kernel<FT, MODE>), dim3(batch, num_heads), dim3(64), 0, s, k_spans, v_spans,
                       q_out, qkv, old_seq_lens, inv_freq, num_heads, num_groups, span_len, span_stride);
  });
  if (!ok) return DIHIP_PARAM_ERROR;
  return launch_status();
}

int dihip_kv_context_copy(void* stream, void* const* spans, const void* src, int src_stride, int seq_len,
                          int start_pos, int num_groups, int head_size, int span_len, int kv_mode, int dtype) {
  DIHIP_REQUIRE(seq_len >= 0 && num_groups > 0 && start_pos >= 0, DIHIP_PARAM_ERROR, "kv_context_copy: bad shape");
  DIHIP_REQUIRE(span_len_ok(span_len), DIHIP_PARAM_ERROR, "kv_context_copy: span size %d", span_len);
  DIHIP_REQUIRE(spans && src, DIHIP_PARAM_ERROR, "kv_context_copy: null pointer");
  if (seq_len == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool ok = kv_dispatch(dtype, kv_mode, head_size, [&](auto ftc, auto mc, auto ec) {
    constexpr int FT = decltype(ftc)::value, MODE = decltype(mc)::value, EPL = decltype(ec)::value;
    hipLaunchKernelGGL((kv_context_copy_kernel<FT, MODE, EPL>), dim3(seq_len, num_groups), dim3(64), 0, s, spans, src,
                       src_stride, start_pos, num_groups, span_len);
  });
  if (!ok) return DIHIP_PARAM_ERROR;
  return launch_status();
}

int dihip_kv_prefix_gather(void* stream, void* dst, void* const* spans, int prefix_len, int num_groups, int head_size,
                           int span_len, int kv_mode, int dtype) {
  DIHIP_REQUIRE(prefix_len >= 0 && num_groups > 0, DIHIP_PARAM_ERROR, "kv_prefix_gather: bad shape");
  DIHIP_REQUIRE(span_len_ok(span_len), DIHIP_PARAM_ERROR, "kv_prefix_gather: span size %d", span_len);
  DIHIP_REQUIRE(dst && spans, DIHIP_PARAM_ERROR, "kv_prefix_gather: null pointer");
  if (prefix_len == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool ok = kv_dispatch(dtype, kv_mode, head_size, [&](auto ftc, auto mc, auto ec) {
    constexpr int FT = decltype(ftc)::value, MODE = decltype(mc)::value, EPL = decltype(ec)::value;
    hipLaunchKernelGGL((kv_prefix_gather_kernel<FT, MODE, EPL>), dim3(prefix_len, num_groups), dim3(64), 0, s, dst,
                       spans, num_groups, span_len);
  });
  if (!ok) return DIHIP_PARAM_ERROR;
  return launch_status();
}

int dihip_span_gather(void* stream, void* dst_cont, void* const* spans, int num_spans, size_t span_bytes) {
  DIHIP_REQUIRE(num_spans >= 0 && dst_cont && spans, DIHIP_PARAM_ERROR, "span_gather: bad argument");
  if (num_spans == 0 || span_bytes == 0) return DIHIP_SUCCESS;
  const int bx = (int)std::min<size_t>(64, (span_bytes / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(span_mass_copy_kernel<true>, dim3(bx, num_spans), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (unsigned char*)dst_cont, spans, span_bytes);
  return launch_status();
}

int dihip_span_scatter(void* stream, void* const* spans, const void* src_cont, int num_spans, size_t span_bytes) {
  DIHIP_REQUIRE(num_spans >= 0 && src_cont && spans, DIHIP_PARAM_ERROR, "span_scatter: bad argument");
  if (num_spans == 0 || span_bytes == 0) return DIHIP_SUCCESS;
  const int bx = (int)std::min<size_t>(64, (span_bytes / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(span_mass_copy_kernel<false>, dim3(bx, num_spans), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (unsigned char*)const_cast<void*>(src_cont), spans,
                     span_bytes);
  return launch_status();
}

}  // extern "C"

namespace dihip {
int rope_table_kv_append(void* stream, void* const* k_spans, void* const* v_spans, void* q_out, const void* qkv,
                         const uint32_t* old_seq_lens, const float* rope_table, int batch, int num_heads, int num_groups,
                         int span_len, int span_stride, int kv_mode, int dtype) {
  DIHIP_REQUIRE(k_spans && v_spans && q_out && qkv && old_seq_lens && rope_table, DIHIP_PARAM_ERROR, "rope_table_kv_append: null pointer");
  if (batch == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool ok = kv_dispatch(dtype, kv_mode, 128, [&](auto ftc, auto mc, auto ec) {
    constexpr int FT = decltype(ftc)::value, MODE = decltype(mc)::value;
    (void)ec;
    hipLaunchKernelGGL((rope_kv_append_kernel<FT, MODE>), dim3(batch, num_heads), dim3(64), 0, s, k_spans, v_spans, q_out, qkv,
                       old_seq_lens, static_cast<const float*>(nullptr), num_heads, num_groups, span_len, span_stride, rope_table);
  });
  if (!ok) return DIHIP_PARAM_ERROR;
  return launch_status();
}
}  // namespace dihip
