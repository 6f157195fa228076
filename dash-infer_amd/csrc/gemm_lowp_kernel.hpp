// gemm_lowp_kernel.hpp -- weight-only (A16W8 / A16W4) GEMM/GEMV for gfx950.
//
// Replaces the reference's whole A16W8/A16W4 kernel family
// (csrc/core/kernel/cuda/gemm_lowp/gemm_a16w8_{perc,subc}_kernel.cu,
//  gemm_a16w4_{perc,subc}_kernel.cu; selection gemm_a16w8_kernel.h:239-331) with ONE
// MI355X-first design that serves M = 1 (decode GEMV), small-M batched decode and large-M
// prefill with the same, batch-invariant numerics:
//
//   * Weights are streamed from HBM exactly once, in "dihip tile-major" order (pack kernels in
//     gemm_lowp.hip): one 16-byte load per lane is already the B fragment of
//     v_mfma_f32_16x16x32_{bf16,f16} (lane = kb*16 + n%16 holds 8 consecutive k per dword), so a
//     wave-load is a fully contiguous 1 KiB and needs no LDS round trip and no shuffles.
//   * The integers are NOT dequantised per element.  q is expanded to the activation type with
//     a one-op magic-number trick (bf16: 0x4300|q = 128+q ; f16: 0x6400|q = 1024+q) and fed to
//     the MFMA as exact small integers; the per-(group, column) scale / zero-point is applied
//     once per group to the f32 accumulator:
//         y[m,n] += s[g,n] * ( sum_k x[m,k] * q'[k,n]  -  (z[g,n] + OFFSET) * sum_k x[m,k] )
//     with sum_k x[m,k] produced by one extra MFMA against an all-ones B fragment.  Products
//     are exact, accumulation is f32: closer to the f32 CPU oracle than the reference's
//     bf16-arithmetic dequantisation.
//   * Activations (<= 32 rows per block) are staged once per block in LDS as the A operand;
//     rows >= M alias a zero row, so M = 1 costs the same instruction stream as M = 16.
//   * Split-K across workgroups with an in-launch, deterministic last-arriver reduction
//     (agent-scope release/acquire + ticket counter, cdna_hip_programming.md G16): slabs are
//     summed in split order, never with float atomics, so results are bit-reproducible.
//   * Epilogues are fused: alpha, bias, UnaryType activation, residual add (EPI_STD), SwiGLU over
//     a gate/up weight pair (EPI_SWIGLU), f32 hidden-stream update (EPI_ADDTO); the RMSNorm of
//     the f32 hidden stream can be fused as prologue (PRO_RMSNORM) for M <= 4.
#pragma once
#include "device_utils.h"

namespace dihip {

enum { PRO_PLAIN = 0, PRO_RMSNORM = 1 };
enum { EPI_STD = 0, EPI_SWIGLU = 1, EPI_ADDTO = 2 };

constexpr int GEMM_THREADS = 256;
constexpr int GEMM_WAVES = 4;
constexpr int GEMM_LDS_HEADER = 1024;  // flag word + reduction scratch (bytes)
constexpr int GEMM_RING = 4;           // k-tiles in flight per column tile per wave

struct GemmArgs {
  const u32x4_t* w0;
  const u32x4_t* w1;  // second weight (EPI_SWIGLU: "up"), else unused
  const uint32_t* sz0;
  const uint32_t* sz1;
  const void* x;  // PRO_PLAIN: FT [M, ldx]; PRO_RMSNORM: f32 [M, ldx]
  int ldx;
  const void* gamma;  // PRO_RMSNORM: FT [K]
  float eps;
  float* slabs;        // [splitk][M][slab_cols] f32
  unsigned* counters;  // [col_blocks * m_blocks], zero on entry, zero on exit
  const void* bias;      // FT [N] or null
  const void* residual;  // FT [M, ldy] or null (EPI_STD)
  void* y;               // FT [M, ldy]
  int ldy;
  const float* h_res;  // EPI_ADDTO: f32 [M, N]
  float* h_out;        // EPI_ADDTO: f32 [M, N]
  float alpha;
  int act;
  int M, N, K;
  int Np;      // N padded to 16
  int KT;      // number of k-tiles (128 k for W4, 64 k for W8)
  int NTILES;  // Np / 16
  int ksteps_per_group;  // group_size / 32, or a huge number for per-channel
  int Gp;      // (scale, zero) groups stored per column tile: sz layout [NTILES][Gp][16]
  int splitk;
  int ktiles_per_split;
  int kslice_tiles;  // k-tiles staged in LDS at a time (== ktiles_per_split for decode shapes)
};

template <int WBITS>
struct WTraits {
  // 32-wide k-steps per 16-byte lane chunk (W16 = unquantised bf16/f16 weights, e.g. lm_head)
  static constexpr int KSTEPS = WBITS == 4 ? 4 : WBITS == 8 ? 2 : 1;
  static constexpr int KTILE = 32 * KSTEPS;
};

// ---- integer -> activation-type expansion (one B fragment = 8 k-values of one column) -----
template <int WBITS, int FT>
struct Expand;

// W4: the dword holds nibble j at bit 4*(j/2) + 16*(j%2), so one shift brings the pair (2i, 2i+1)
// to the TOP four mantissa bits of both 16-bit halves; OR-ing the exponent of 16.0 yields 16+q
// exactly (bf16: mantissa bits 3..6 of [16,32) count units; f16: bits 6..9).  The small offset
// (16, merged into the zero point) keeps the f32 accumulation noise two orders below one FT ulp.
template <int FT>
struct Expand<4, FT> {
  static constexpr float OFFSET = 16.f;
  static constexpr int POS = FT == DIHIP_BF16 ? 3 : 6;  // bit position of the nibble LSB
  static constexpr uint32_t MASK = 0x000F000Fu << POS;
  static constexpr uint32_t MAGIC = FT == DIHIP_BF16 ? 0x41804180u : 0x4C004C00u;
  template <int I>
  __device__ __forceinline__ static uint32_t pair(uint32_t d) {
    constexpr int sh = 4 * I - POS;  // > 0: shift right, < 0: shift left
    const uint32_t v = sh >= 0 ? (d >> (sh >= 0 ? sh : 0)) : (d << (sh < 0 ? -sh : 0));
    return (v & MASK) | MAGIC;
  }
  __device__ __forceinline__ static u32x4_t frag(const u32x4_t& chunk, int ks) {
    const uint32_t d = chunk[ks];
    return u32x4_t{pair<0>(d), pair<1>(d), pair<2>(d), pair<3>(d)};
  }
};
// W8: bytes are u8 = q + 128 in natural k order.  f16: 0x6400|u8 = 1024+u8 exact.
// bf16 has only 7 mantissa bits so the magic OR cannot cover 8 bits: convert through f32
// (v_cvt_f32_ubyteN) and keep the (exact) upper halves.
template <>
struct Expand<8, DIHIP_F16> {
  static constexpr float OFFSET = 1024.f + 128.f;
  __device__ __forceinline__ static u32x4_t frag(const u32x4_t& chunk, int ks) {
    const uint32_t d0 = chunk[2 * ks], d1 = chunk[2 * ks + 1];
    u32x4_t r;
    r[0] = ((d0 & 0xFFu) | ((d0 & 0xFF00u) << 8)) | 0x64006400u;
    r[1] = (((d0 >> 16) & 0xFFu) | ((d0 >> 8) & 0xFF0000u)) | 0x64006400u;
    r[2] = ((d1 & 0xFFu) | ((d1 & 0xFF00u) << 8)) | 0x64006400u;
    r[3] = (((d1 >> 16) & 0xFFu) | ((d1 >> 8) & 0xFF0000u)) | 0x64006400u;
    return r;
  }
};
template <>
struct Expand<8, DIHIP_BF16> {
  static constexpr float OFFSET = 128.f;
  __device__ __forceinline__ static uint32_t pair(uint32_t lo_byte, uint32_t hi_byte) {
    const uint32_t f0 = __float_as_uint((float)lo_byte);
    const uint32_t f1 = __float_as_uint((float)hi_byte);
    return (f0 >> 16) | (f1 & 0xFFFF0000u);
  }
  __device__ __forceinline__ static u32x4_t frag(const u32x4_t& chunk, int ks) {
    const uint32_t d0 = chunk[2 * ks], d1 = chunk[2 * ks + 1];
    u32x4_t r;
    r[0] = pair(d0 & 0xFFu, (d0 >> 8) & 0xFFu);
    r[1] = pair((d0 >> 16) & 0xFFu, d0 >> 24);
    r[2] = pair(d1 & 0xFFu, (d1 >> 8) & 0xFFu);
    r[3] = pair((d1 >> 16) & 0xFFu, d1 >> 24);
    return r;
  }
};

// W16: the chunk already is the fragment (8 consecutive k of one column in FT)
template <int FT>
struct Expand<16, FT> {
  static constexpr float OFFSET = 0.f;
  __device__ __forceinline__ static u32x4_t frag(const u32x4_t& chunk, int) { return chunk; }
};

template <int FT>
__device__ __forceinline__ f32x4_t mfma16(const u32x4_t& a, const u32x4_t& b, const f32x4_t& c) {
  if constexpr (FT == DIHIP_BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
}

// One output element (m, n): v = k-sum of the first weight, v2 = of the second (EPI_SWIGLU).
template <int FT, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, int m, int n, float v, float v2) {
  if constexpr (EPI == EPI_STD) {
    // reduce_sum order of the reference (gemm_a16w8_subc_kernel.cu:953-977):
    // act(alpha * sum + bias); then the Gemm op's fused binary ADD.
    v = a.alpha * v;
    if (a.bias) v += load_ft<FT>(a.bias, n);
    v = apply_act(v, a.act);
    if (a.residual) v = ft_round<FT>(v) + load_ft<FT>(a.residual, (size_t)m * a.ldy + n);
    store_ft<FT>(a.y, (size_t)m * a.ldy + n, v);
  } else if constexpr (EPI == EPI_SWIGLU) {
    // Gemm(gate, SiLU) x Gemm(up) -> Binary MUL (qwen_v15.py:314-335), one rounding
    store_ft<FT>(a.y, (size_t)m * a.ldy + n, (v / (1.f + expf(-v))) * v2);
  } else {
    // f32 hidden-stream update; h_res == nullptr -> plain f32 output (lm_head logits, TP partials)
    const float base = a.h_res ? a.h_res[(size_t)m * a.N + n] : 0.f;
    a.h_out[(size_t)m * a.N + n] = base + a.alpha * v;
  }
}

// MT : 16-row m-tiles per block (1: M<=16, 2: M<=32 per grid.z slice)
// NT : 16-column n-tiles per wave (EPI_SWIGLU: NT/2 gate tiles + NT/2 up tiles, same columns)
template <int WBITS, int FT, int MT, int NT, int PRO, int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_lowp_kernel(const GemmArgs a) {
  using WT = WTraits<WBITS>;
  using EX = Expand<WBITS, FT>;
  constexpr int KSTEPS = WT::KSTEPS;
  constexpr int KTILE = WT::KTILE;
  constexpr int NTW = EPI == EPI_SWIGLU ? NT / 2 : NT;  // distinct column tiles per wave
  constexpr int ROWS = 16 * MT;
  constexpr bool QUANT = WBITS != 16;
  // k-tiles of every column tile kept in flight per wave (register ring).  Decode-sized
  // launches are latency bound: the whole per-wave K range should be in flight at once.
  constexpr int D = GEMM_RING;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* flag_lds = reinterpret_cast<unsigned*>(smem);
  float* red = reinterpret_cast<float*>(smem + 16);
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem + GEMM_LDS_HEADER);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int ni = lane & 15, kb = lane >> 4;
  const int cb = blockIdx.x, split = blockIdx.y, mz = blockIdx.z;
  const int m0 = mz * ROWS;
  const int rows = min(ROWS, a.M - m0);

  const int kt0 = split * a.ktiles_per_split;
  const int kt1 = min(a.KT, kt0 + a.ktiles_per_split);
  const int RS = a.kslice_tiles * KTILE + 8;  // LDS row stride in elements (16-byte pad)

  const int tile0 = (cb * GEMM_WAVES + wave) * NTW;  // first column tile of this wave
  const bool wave_active = tile0 < a.NTILES;
  const int last_grp = (a.KT * KSTEPS - 1) / a.ksteps_per_group;
  const bool small_groups = QUANT && a.ksteps_per_group < KSTEPS;  // several groups per k-tile

  // ---- weight / (scale, zero) register ring; first D k-tiles are issued before touching x ----
  u32x4_t wbuf[D][NT];
  uint32_t szbuf[D][NT];
  const u32x4_t* wbase[NT];
  const uint32_t* szbase[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tile = min(tile0 + (t % NTW), a.NTILES - 1);
    const bool second = EPI == EPI_SWIGLU && t >= NTW;
    wbase[t] = (second ? a.w1 : a.w0) + (size_t)tile * a.KT * 64 + lane;
    szbase[t] = (second ? a.sz1 : a.sz0) + (size_t)tile * a.Gp * 16 + ni;
  }
#define DIHIP_ISSUE(SLOT, KT_)                                                                    \
  do {                                                                                            \
    _Pragma("unroll") for (int t_ = 0; t_ < NT; ++t_) {                                           \
      wbuf[SLOT][t_] = __builtin_nontemporal_load(wbase[t_] + (size_t)(KT_) * 64);                \
      if constexpr (QUANT)                                                                        \
        szbuf[SLOT][t_] = szbase[t_][(size_t)min(((KT_) * KSTEPS) / a.ksteps_per_group, last_grp) * 16]; \
    }                                                                                             \
  } while (0)
  if (wave_active) {
#pragma unroll
    for (int j = 0; j < D; ++j)
      if (kt0 + j < kt1) DIHIP_ISSUE(j, kt0 + j);
  }

  if constexpr (PRO == PRO_RMSNORM) {
    // RMSNorm (LayerNormNoBeta, csrc/core/kernel/cpu/layernorm.cpp:110-157) of the f32 hidden
    // stream: rstd = 1/sqrt(mean(x^2)+eps); x_norm = FT((gamma*x)*rstd).  rows <= 4 here.
    const float* hg = reinterpret_cast<const float*>(a.x);
    for (int r = 0; r < rows; ++r) {
      float ss = 0.f;
      const float* hr = hg + (size_t)(m0 + r) * a.ldx;
      for (int k = tid * 4; k < a.K; k += GEMM_THREADS * 4) {
        if (k + 4 <= a.K && ((a.ldx & 3) == 0)) {
          const f32x4_t v = *reinterpret_cast<const f32x4_t*>(hr + k);
          ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        } else {
          for (int j = 0; j < 4 && k + j < a.K; ++j) ss += hr[k + j] * hr[k + j];
        }
      }
      ss = wave_sum(ss);
      if (lane == 0) red[r * GEMM_WAVES + wave] = ss;
    }
  }
  // zero row (index `rows`) read by every lane whose A row is >= M
  for (int i = tid; i < RS; i += GEMM_THREADS) xs[(size_t)rows * RS + i] = 0;

  // per-lane A row offsets (in elements); rows beyond M read the zero row
  int arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = mt * 16 + ni;
    arow[mt] = (r < rows ? r : rows) * RS + kb * 8;
  }

  f32x4_t tot[MT][NT], gacc[MT][NT], xsum[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    xsum[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      tot[mt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      gacc[mt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  const u32x4_t ones = FT == DIHIP_BF16 ? u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}
                                        : u32x4_t{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};

  const int kstep0 = kt0 * KSTEPS;
  int grp = kstep0 / a.ksteps_per_group;
  int ksg = kstep0 - grp * a.ksteps_per_group;  // k-steps already consumed of the current group

  // s * (acc - (z + OFFSET) * xsum) of the finished group into the running total
  auto fixup = [&](const uint32_t (&szv)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float s = ft_bits_to_f32<FT>(szv[t] & 0xFFFFu);
      const float zp = ft_bits_to_f32<FT>(szv[t] >> 16) + EX::OFFSET;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          tot[mt][t][r] = fmaf(s, fmaf(-zp, xsum[mt][r], gacc[mt][t][r]), tot[mt][t][r]);
          gacc[mt][t][r] = 0.f;
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xsum[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  uint32_t sz_last[NT];  // (scale, zero) of the group the most recent k-tile belongs to
#pragma unroll
  for (int t = 0; t < NT; ++t) sz_last[t] = 0;

  // ---- main loop over LDS-staged k-slices (one slice for decode shapes) ---------------------
  for (int s0 = kt0; s0 < kt1; s0 += a.kslice_tiles) {
    const int s1 = min(kt1, s0 + a.kslice_tiles);
    const int k0 = s0 * KTILE;
    const int krange = (s1 - s0) * KTILE;
    __syncthreads();  // previous slice consumed / rmsnorm partials and zero row visible
    if constexpr (PRO == PRO_PLAIN) {
      const uint16_t* xg = reinterpret_cast<const uint16_t*>(a.x);
      const bool vec_ok = ((a.ldx & 7) == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
      const int nvec = krange >> 3;
      for (int i = tid; i < rows * nvec; i += GEMM_THREADS) {
        const int r = i / nvec, c = (i - r * nvec) << 3;
        const int k = k0 + c;
        u32x4_t v = {0, 0, 0, 0};
        const uint16_t* src = xg + (size_t)(m0 + r) * a.ldx + k;
        if (vec_ok && k + 8 <= a.K) {
          v = *reinterpret_cast<const u32x4_t*>(src);
        } else {
          uint16_t tmp[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) tmp[j] = (k + j < a.K) ? src[j] : (uint16_t)0;
          v = u32x4_t{(uint32_t)tmp[0] | ((uint32_t)tmp[1] << 16), (uint32_t)tmp[2] | ((uint32_t)tmp[3] << 16),
                      (uint32_t)tmp[4] | ((uint32_t)tmp[5] << 16), (uint32_t)tmp[6] | ((uint32_t)tmp[7] << 16)};
        }
        *reinterpret_cast<u32x4_t*>(xs + (size_t)r * RS + c) = v;
      }
    } else {
      const float* hg = reinterpret_cast<const float*>(a.x);
      for (int i = tid; i < rows * krange; i += GEMM_THREADS) {
        const int r = i / krange, c = i - r * krange;
        const int k = k0 + c;
        float v = 0.f;
        if (k < a.K) {
          const float ss = red[r * GEMM_WAVES] + red[r * GEMM_WAVES + 1] + red[r * GEMM_WAVES + 2] +
                           red[r * GEMM_WAVES + 3];
          const float rstd = 1.f / sqrtf(ss / (float)a.K + a.eps);
          const float g = load_ft<FT>(a.gamma, k);
          v = (g * hg[(size_t)(m0 + r) * a.ldx + k]) * rstd;
        }
        xs[(size_t)r * RS + c] = (uint16_t)f32_to_ft_bits<FT>(v);
      }
    }
    __syncthreads();

    if (wave_active) {
      // (s0 - kt0) is a multiple of D (host plan), so ring slot j always holds k-tile base + j
      for (int base = s0; base < s1; base += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const int kt = base + j;
          if (kt < s1) {  // wave-uniform
            const int kl = (kt - s0) * KTILE;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
              u32x4_t af[MT];
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
                af[mt] = *reinterpret_cast<const u32x4_t*>(xs + arow[mt] + kl + ks * 32);
                if constexpr (QUANT) xsum[mt] = mfma16<FT>(af[mt], ones, xsum[mt]);
              }
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const u32x4_t bf = EX::frag(wbuf[j][t], ks);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                  if constexpr (QUANT) gacc[mt][t] = mfma16<FT>(af[mt], bf, gacc[mt][t]);
                  else tot[mt][t] = mfma16<FT>(af[mt], bf, tot[mt][t]);
                }
              }
              if constexpr (QUANT) {
                if (++ksg == a.ksteps_per_group) {  // wave-uniform
                  if (small_groups) {  // rare: group < k-tile, fetch its parameters on demand
                    uint32_t szv[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) szv[t] = szbase[t][(size_t)min(grp, last_grp) * 16];
                    fixup(szv);
                  } else {
                    fixup(szbuf[j]);
                  }
                  ksg = 0;
                  ++grp;
                }
              }
            }
            if constexpr (QUANT) {
#pragma unroll
              for (int t = 0; t < NT; ++t) sz_last[t] = szbuf[j][t];
            }
            if (kt + D < kt1) DIHIP_ISSUE(j, kt + D);
          }
        }
      }
    }
  }
  if constexpr (QUANT) {
    if (wave_active && ksg != 0) {  // per-channel, or a K range that ends inside a group
      if (small_groups) {
#pragma unroll
        for (int t = 0; t < NT; ++t) sz_last[t] = szbase[t][(size_t)min(grp, last_grp) * 16];
      }
      fixup(sz_last);
    }
  }
#undef DIHIP_ISSUE

  // ---- no split: epilogue straight from the accumulator registers ----------------------------
  // C layout of v_mfma_f32_16x16x32: lane holds column ni, rows kb*4 + r.
  if (a.splitk == 1) {
    if (!wave_active) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mt * 16 + kb * 4 + r;
        if (m >= a.M) continue;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const int n = (tile0 + t) * 16 + ni;
          if (n >= a.N) continue;
          gemm_epilogue<FT, EPI>(a, m, n, tot[mt][t][r], EPI == EPI_SWIGLU ? tot[mt][(t + NTW) % NT][r] : 0.f);
        }
      }
    return;
  }

  // ---- split-K hand-off: slabs [split][M][slab_cols] f32, last arriver reduces ----------------
  const int slab_cols = (EPI == EPI_SWIGLU ? 2 : 1) * a.Np;
  if (wave_active) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + mt * 16 + kb * 4 + r;
        if (m < a.M) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int col = ((EPI == EPI_SWIGLU && t >= NTW) ? a.Np : 0) + (tile0 + (t % NTW)) * 16 + ni;
            if (tile0 + (t % NTW) < a.NTILES) a.slabs[((size_t)split * a.M + m) * slab_cols + col] = tot[mt][t][r];
          }
        }
      }
  }
  unsigned* counter = a.counters + (size_t)mz * gridDim.x + cb;
  if (!arrive_and_check_last(counter, (unsigned)a.splitk, flag_lds)) return;

  // Last arriver: every thread owns whole output elements; the split partials of an element are
  // fetched in independent batches (all loads of a batch in flight together) and summed in split
  // order, so the result does not depend on arrival order.
  constexpr int BLOCK_COLS = GEMM_WAVES * NTW * 16;
  const int ncols = min(BLOCK_COLS, a.Np - cb * BLOCK_COLS);
  for (int e = tid; e < rows * ncols; e += GEMM_THREADS) {
    const int mr = e / ncols, c = e - mr * ncols;
    const int m = m0 + mr, n = cb * BLOCK_COLS + c;
    if (n >= a.N) continue;
    const float* p = a.slabs + (size_t)m * slab_cols + n;
    const size_t sstride = (size_t)a.M * slab_cols;
    float acc = 0.f, acc2 = 0.f;
    for (int sb = 0; sb < a.splitk; sb += 16) {
      float v[16], v2[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const bool in = sb + j < a.splitk;
        v[j] = in ? p[(size_t)(sb + j) * sstride] : 0.f;
        if constexpr (EPI == EPI_SWIGLU) v2[j] = in ? p[(size_t)(sb + j) * sstride + a.Np] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc += v[j];
        if constexpr (EPI == EPI_SWIGLU) acc2 += v2[j];
      }
    }
    gemm_epilogue<FT, EPI>(a, m, n, acc, acc2);
  }
}

}  // namespace dihip
