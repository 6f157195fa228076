// span_codec.hpp -- the KV span codec on the device: quantise + store / load + dequantise one token-head of a span
// (byte layout and arithmetic: span_cache.hip's header; impl_i8.cuh:53-66,116-142, impl_u4.cuh:79-103,157-184).  Shared by the span
// writers (span_cache.hip) and the decode-step attention that appends the new token itself (span_attn.hip).
#pragma once
#include "device_utils.h"

namespace dihip {

template <int FT>
struct FtBytes {
  static constexpr int v = FT == DIHIP_F32 ? 4 : 2;
};

__host__ __device__ inline size_t span_data_bytes(int g, int S, int H, int mode, int ft) {
  if (mode == DIHIP_KV_NONE) return (size_t)g * S * H * (ft == DIHIP_F32 ? 4 : 2);
  if (mode == DIHIP_KV_I8) return (size_t)g * S * H;
  return (size_t)g * S * H / 2;
}

// Quantise + store one token-head.  x[EPL] are this lane's elements (d = lane*EPL + i).
template <int FT, int MODE, int EPL>
__device__ __forceinline__ void store_token_head(void* span, const float (&x)[EPL], int head, int pos, int g, int S,
                                                 int H, int lane) {
  if constexpr (MODE == DIHIP_KV_NONE) {
    const size_t base = ((size_t)head * S + pos) * H + lane * EPL;
#pragma unroll
    for (int i = 0; i < EPL; ++i) store_ft<FT>(span, base + i, x[i]);
  } else {
    constexpr float QMAX = MODE == DIHIP_KV_I8 ? 127.f : 15.f;
    constexpr float QMIN = MODE == DIHIP_KV_I8 ? -128.f : 0.f;
    float mx = x[0], mn = x[0];
#pragma unroll
    for (int i = 1; i < EPL; ++i) {
      mx = fmaxf(mx, x[i]);
      mn = fminf(mn, x[i]);
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    float qs = (mx - mn) / (QMAX - QMIN);
    qs = fmaxf(qs, 1e-5f);
    float qz = QMIN - mn / qs;
    qz = fminf(qz, QMAX);
    if constexpr (MODE == DIHIP_KV_I8) qz = fmaxf(qz, QMIN);
    qz = rintf(qz);
    const int HB = MODE == DIHIP_KV_I8 ? H : H / 2;
    unsigned char* data = reinterpret_cast<unsigned char*>(span) + ((size_t)head * S + pos) * HB;
    if constexpr (MODE == DIHIP_KV_I8) {
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        float t = qz + x[i] / qs;
        t = fminf(t, QMAX);
        t = fmaxf(t, QMIN);
        data[lane * EPL + i] = (unsigned char)(signed char)rintf(t);
      }
    } else {
      static_assert(MODE != DIHIP_KV_U4 || EPL % 2 == 0, "u4 packs two elements per byte");
#pragma unroll
      for (int i = 0; i < EPL; i += 2) {
        // impl_u4.cuh:79-93: min(., 15), rint, static_cast<uint32_t> -- the float -> u32 convert
        // SATURATES (negatives -> 0; reachable when zero clamps at 15 on an all-negative head)
        float t0 = fmaxf(rintf(fminf(qz + x[i] / qs, QMAX)), 0.f);
        float t1 = fmaxf(rintf(fminf(qz + x[i + 1] / qs, QMAX)), 0.f);
        const unsigned w0 = (unsigned)t0 & 0xFu, w1 = (unsigned)t1 & 0xFu;
        data[(lane * EPL + i) >> 1] = (unsigned char)(w0 | (w1 << 4));
      }
    }
    if (lane == 0) {
      float* params = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(span) + (size_t)g * S * HB) +
                      ((size_t)head * S + pos) * 2;
      params[0] = qz;
      params[1] = qs;
    }
  }
}

// Load + dequantise this lane's EPL elements of one token-head.
template <int FT, int MODE, int EPL>
__device__ __forceinline__ void load_token_head(const void* span, float (&x)[EPL], int head, int pos, int g, int S,
                                                int H, int lane) {
  if constexpr (MODE == DIHIP_KV_NONE) {
    const size_t base = ((size_t)head * S + pos) * H + lane * EPL;
#pragma unroll
    for (int i = 0; i < EPL; ++i) x[i] = load_ft<FT>(span, base + i);
  } else {
    const int HB = MODE == DIHIP_KV_I8 ? H : H / 2;
    const unsigned char* data = reinterpret_cast<const unsigned char*>(span) + ((size_t)head * S + pos) * HB;
    const float* params =
        reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(span) + (size_t)g * S * HB) +
        ((size_t)head * S + pos) * 2;
    const float zero = params[0], scale = params[1];
    if constexpr (MODE == DIHIP_KV_I8) {
#pragma unroll
      for (int i = 0; i < EPL; ++i) x[i] = ((float)(signed char)data[lane * EPL + i] - zero) * scale;
    } else {
#pragma unroll
      for (int i = 0; i < EPL; i += 2) {
        const unsigned b = data[(lane * EPL + i) >> 1];
        x[i] = ((float)(b & 0xFu) - zero) * scale;
        x[i + 1] = ((float)(b >> 4) - zero) * scale;
      }
    }
  }
}

}  // namespace dihip
