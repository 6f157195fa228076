/* oracle/c/oracle_c.c -- plain-C restatement of the reference's CPU semantics for the hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle_c.h / oracle/__init__.py).  Compiled with
 * -ffp-contract=off so every f32 operation rounds exactly like the reference's host loops. */
#include "oracle_c.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ numerics ------------ */
static float bf16_round(float x) { /* RNE, csrc/common/bfloat16_impl.hpp */
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN */
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  memcpy(&x, &u, 4);
  return x;
}
static uint16_t f32_to_f16_bits(float x) { /* IEEE binary16, round-to-nearest-even */
  uint32_t u;
  memcpy(&u, &x, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0)); /* inf/nan */
  if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                  /* overflow */
  if (u < 0x38800000u) { /* subnormal half or zero */
    if (u < 0x33000000u) return (uint16_t)sign;
    const int e = (int)(u >> 23);
    uint32_t m = (u & 0x7fffffu) | 0x800000u;
    const int shift = 126 - e; /* 14..24 */
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = (u - 0x38000000u) >> 13;
  const uint32_t rem = u & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
  return (uint16_t)(sign | r);
}
static float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
  if (e == 0) {
    if (m == 0) u = sign;
    else {
      int s = 0;
      while (!(m & 0x400)) { m <<= 1; s++; }
      u = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ff) << 13);
    }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static float f16_round(float x) { return f16_bits_to_f32(f32_to_f16_bits(x)); }
float orc_round_ft(float x, int ft) {
  return ft == ORC_BF16 ? bf16_round(x) : ft == ORC_F16 ? f16_round(x) : x;
}
static uint16_t ft_bits16(float x, int ft) {
  if (ft == ORC_BF16) {
    float r = bf16_round(x);
    uint32_t u;
    memcpy(&u, &r, 4);
    return (uint16_t)(u >> 16);
  }
  return f32_to_f16_bits(x);
}
static float ft_from_bits16(uint16_t b, int ft) {
  if (ft == ORC_BF16) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  }
  return f16_bits_to_f32(b);
}

/* ------------------------------------------------------------------ gemm ---------------- */
static inline float wq_at(const void* B, int k, int n, int N, int wbits) {
  if (wbits == 8) return (float)((const int8_t*)B)[(size_t)k * N + n];
  const int NP = (N + 1) / 2;
  uint8_t b = ((const uint8_t*)B)[(size_t)k * NP + n / 2];
  return (float)((n & 1) ? (b >> 4) : (b & 0xf)); /* convert_4bit.h:9-16 */
}

static float apply_act(float v, int act) {
  switch (act) { /* UnaryType values, csrc/proto/allspark.proto:68-76 */
    case 1: return tanhf(v);
    case 2: return 0.5f * v * (1.f + erff(v * 0.70710678f));
    case 3: return 0.5f * v * (1.f + tanhf(0.7978845608f * (v + 0.044715f * v * v * v)));
    case 4: return v > 0.f ? v : 0.f;
    case 5: return v / (1.f + expf(-v));
    case 6: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

int orc_gemm_a16wx(const float* A, const void* B, const float* S, const float* Z,
                   const float* bias, float* C, int M, int N, int K, int group, int wbits,
                   float alpha, int act, int ft) {
  const int G = group > 0 ? group : K;
#pragma omp parallel for collapse(2)
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float sum = 0.f;
      for (int k = 0; k < K; ++k) {
        const int gi = k / G;
        float tmp = (wq_at(B, k, n, N, wbits) - Z[(size_t)gi * N + n]) * S[(size_t)gi * N + n];
        sum += A[(size_t)m * K + k] * tmp;
      }
      float v = alpha * sum;
      if (bias) v += bias[n];
      v = apply_act(v, act);
      C[(size_t)m * N + n] = orc_round_ft(v, ft);
    }
  return 0;
}

int orc_gemm_a16wx_x86bf16(const float* A, const void* B, const float* S, const float* Z,
                           const float* bias, float* C, int M, int N, int K, int group,
                           int wbits, float alpha, int act) {
  const int G = group > 0 ? group : K;
#pragma omp parallel for collapse(2)
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float sum = 0.f;
      for (int k = 0; k < K; ++k) {
        const int gi = k / G;
        float w = (wq_at(B, k, n, N, wbits) - Z[(size_t)gi * N + n]) * S[(size_t)gi * N + n];
        sum += bf16_round(A[(size_t)m * K + k]) * bf16_round(w);
      }
      float v = alpha * sum;
      if (bias) v += bias[n];
      C[(size_t)m * N + n] = apply_act(v, act);
    }
  return 0;
}

/* ------------------------------------------------------------------ KV codec ------------ */
static int elem_bytes(int ft) { return ft == ORC_F32 ? 4 : 2; }

size_t orc_span_bytes(int g, int S, int H, int mode, int ft) {
  /* csrc/runtime/cache/virtual_cache.cpp:202-232 (CacheUtils::GetSpanSizeInBytes) */
  if (mode == ORC_KV_NONE) return (size_t)g * S * H * elem_bytes(ft);
  if (mode == ORC_KV_I8) return (size_t)g * S * H + (size_t)2 * S * g * 4;
  return (size_t)g * S * H / 2 + (size_t)2 * S * g * 4;
}

void orc_span_write_head(void* span, const float* x, int head, int pos, int g, int S, int H,
                         int mode, int ft) {
  if (mode == ORC_KV_NONE) {
    if (ft == ORC_F32) {
      memcpy((float*)span + ((size_t)head * S + pos) * H, x, sizeof(float) * H);
    } else {
      uint16_t* dst = (uint16_t*)span + ((size_t)head * S + pos) * H;
      for (int d = 0; d < H; ++d) dst[d] = ft_bits16(x[d], ft);
    }
    return;
  }
  /* builder: impl_i8.cuh:116-142 / impl_u4.cuh:157-184.  The values quantised are the FT inputs
   * converted to f32. */
  const float QMAX = mode == ORC_KV_I8 ? 127.f : 15.f;
  const float QMIN = mode == ORC_KV_I8 ? -128.f : 0.f;
  const float RANGE = QMAX - QMIN, ORIGIN = QMIN, EPS = 1e-5f;
  float mx = -INFINITY, mn = INFINITY;
  for (int d = 0; d < H; ++d) {
    float v = orc_round_ft(x[d], ft);
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
  }
  float qs = (mx - mn) / RANGE;
  qs = fmaxf(qs, EPS);
  float qz = ORIGIN - mn / qs;
  qz = fminf(qz, QMAX);
  if (mode == ORC_KV_I8) qz = fmaxf(qz, QMIN); /* u4: "uint naturally >= 0", no lower clamp */
  qz = rintf(qz);                              /* CONFIG_CACHE_ROUND_RNI (config.cuh:13) */

  const int HB = mode == ORC_KV_I8 ? H : H / 2;
  uint8_t* data = (uint8_t*)span + ((size_t)head * S + pos) * HB;
  float* params = (float*)((uint8_t*)span + (size_t)g * S * HB) + ((size_t)head * S + pos) * 2;
  params[0] = qz; /* struct { CPT zero; CPT scale; } */
  params[1] = qs;
  if (mode == ORC_KV_I8) {
    for (int d = 0; d < H; ++d) {
      float t = qz + orc_round_ft(x[d], ft) / qs; /* Quant(): impl_i8.cuh:53-60 */
      t = fminf(t, QMAX);
      t = fmaxf(t, QMIN);
      t = rintf(t);
      ((int8_t*)data)[d] = (int8_t)t;
    }
  } else {
    for (int d = 0; d < H; d += 2) {
      uint32_t w[2];
      for (int i = 0; i < 2; ++i) {
        float t = qz + orc_round_ft(x[d + i], ft) / qs; /* impl_u4.cuh:79-93 */
        t = fminf(t, QMAX);
        t = rintf(t);
        /* static_cast<uint32_t>(float) on the device saturates: negatives -> 0 (reachable when
           zero clamps at 15 on an all-negative head); a plain C cast would be UB here */
        w[i] = t > 0.f ? (uint32_t)t : 0u;
      }
      data[d / 2] = (uint8_t)((w[0] & 0xf) | ((w[1] & 0xf) << 4)); /* impl_u4.cuh:27-29 */
    }
  }
}

void orc_span_read_head(const void* span, float* x, int head, int pos, int g, int S, int H,
                        int mode, int ft) {
  if (mode == ORC_KV_NONE) {
    if (ft == ORC_F32) {
      memcpy(x, (const float*)span + ((size_t)head * S + pos) * H, sizeof(float) * H);
    } else {
      const uint16_t* src = (const uint16_t*)span + ((size_t)head * S + pos) * H;
      for (int d = 0; d < H; ++d) x[d] = ft_from_bits16(src[d], ft);
    }
    return;
  }
  const int HB = mode == ORC_KV_I8 ? H : H / 2;
  const uint8_t* data = (const uint8_t*)span + ((size_t)head * S + pos) * HB;
  const float* params =
      (const float*)((const uint8_t*)span + (size_t)g * S * HB) + ((size_t)head * S + pos) * 2;
  const float zero = params[0], scale = params[1];
  if (mode == ORC_KV_I8) {
    for (int d = 0; d < H; ++d) x[d] = ((float)((const int8_t*)data)[d] - zero) * scale;
  } else {
    for (int d = 0; d < H; d += 2) {
      x[d] = ((float)(data[d / 2] & 0xf) - zero) * scale;
      x[d + 1] = ((float)(data[d / 2] >> 4) - zero) * scale;
    }
  }
}

/* ------------------------------------------------------------------ attention ----------- */
void orc_span_attn_decode(float* out, const float* q, const void* const* kspans,
                          const void* const* vspans, int len, int n, int g, int H, int S, int mode,
                          int ft, float alpha) {
  /* cpu_dec_single_mqa: score = alpha * q.K^T (sgemm), softmax f32, out = P.V; head h uses KV
   * group h / (n / g) (csrc/core/kernel/cpu/mha.cpp:748-765). */
  const int hpg = n / g;
#pragma omp parallel for
  for (int h = 0; h < n; ++h) {
    const int grp = h / hpg;
    float* score = (float*)malloc(sizeof(float) * (size_t)len);
    float* row = (float*)malloc(sizeof(float) * (size_t)H);
    float mx = -INFINITY;
    for (int t = 0; t < len; ++t) {
      orc_span_read_head(kspans[t / S], row, grp, t % S, g, S, H, mode, ft);
      float s = 0.f;
      for (int d = 0; d < H; ++d) s += q[(size_t)h * H + d] * row[d];
      score[t] = alpha * s;
      mx = fmaxf(mx, score[t]);
    }
    float sum = 0.f;
    for (int t = 0; t < len; ++t) {
      score[t] = expf(score[t] - mx);
      sum += score[t];
    }
    for (int d = 0; d < H; ++d) out[(size_t)h * H + d] = 0.f;
    for (int t = 0; t < len; ++t) {
      orc_span_read_head(vspans[t / S], row, grp, t % S, g, S, H, mode, ft);
      const float p = score[t] / sum;
      for (int d = 0; d < H; ++d) out[(size_t)h * H + d] += p * row[d];
    }
    free(score);
    free(row);
  }
}

void orc_prefill_attn(float* out, const float* q, const float* k, const float* v, int Lq, int Lk,
                      int n, int g, int H, float alpha, int causal) {
  /* tests/cpp/kernel/cuda/kernel_mhaprefill_test.cpp:119-250 generalised to GQA and to a
   * cached prefix (Lk >= Lq): query i sees keys j <= i + (Lk - Lq). */
  const int hpg = n / g;
  const int off = Lk - Lq;
#pragma omp parallel for collapse(2)
  for (int h = 0; h < n; ++h)
    for (int i = 0; i < Lq; ++i) {
      const int grp = h / hpg;
      const int kend = causal ? (i + off + 1) : Lk;
      float* score = (float*)malloc(sizeof(float) * (size_t)Lk);
      float mx = -INFINITY;
      for (int j = 0; j < kend; ++j) {
        float s = 0.f;
        for (int d = 0; d < H; ++d)
          s += alpha * q[((size_t)i * n + h) * H + d] * k[((size_t)j * g + grp) * H + d];
        score[j] = s;
        mx = fmaxf(mx, s);
      }
      float sum = 0.f;
      for (int j = 0; j < kend; ++j) {
        score[j] = expf(score[j] - mx);
        sum += score[j];
      }
      for (int d = 0; d < H; ++d) {
        float acc = 0.f;
        for (int j = 0; j < kend; ++j) acc += (score[j] / sum) * v[((size_t)j * g + grp) * H + d];
        out[((size_t)i * n + h) * H + d] = acc;
      }
      free(score);
    }
}
