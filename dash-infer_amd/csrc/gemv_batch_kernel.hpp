// gemv_batch_kernel.hpp -- weight-only GEMM for small decode batches (1 < M <= 32) on gfx950.
//
// Batched decode (BASELINE configs #3 / #4: M = 32 / 16 rows) streams the same weight bytes as
// batch 1, so it is just as launch-latency sensitive -- but M x K activations (up to 1.2 MB) no
// longer fit in LDS, which the batch-1 kernel (gemv_stream_kernel.hpp) relies on.  This kernel keeps
// the no-inter-workgroup-communication structure and the hand-counted weight ring of that kernel
// and changes where the A operand lives:
//
//   * a workgroup (8 waves) owns ONE 16-column tile (SwiGLU: the gate and the up tile of the same
//     columns); the 8 waves split K and combine through LDS in fixed order;
//   * each wave loads the MFMA A fragments of ITS k-range straight from the activation matrix in
//     L2 (lane (kb, row) <- 16 bytes of x[row][k..k+8]) into registers, PH k-tiles at a time, and
//     reuses them for both half-units; nothing is staged in LDS, no barrier before the epilogue;
//   * sum_k x[m][k] of each k-tile (needed by the zero-point term) comes from one extra MFMA per
//     k-step against an all-ones B fragment -- rows land in the C layout the fix-up needs;
//   * weights are loaded PH k-tiles ("a phase") at a time into one of two alternating register sets
//     -- the next phase is in flight while the current one is multiplied -- with plain loads (the
//     hand-counted asm ring of the batch-1 kernel does not survive this control flow), and are
//     expanded / scaled exactly as in the batch-1 kernel; every row goes through the same arithmetic
//     whatever the batch (bit-reproducible; the batch-1 kernel differs only in the Sum x path).
//
// Replaces hgemm_a16w8_32x128x32_16816_nn_splitk / hgemm_a16w4_subc_32x256x32_16816 + reduce_sum
// for M <= 32 (gemm_a16w8_subc_kernel.cu:1132-1521, gemm_a16w4_subc_kernel.cu:466-815).
#pragma once
#include "gemv_stream_kernel.hpp"

namespace dihip {

constexpr int GEMB_WAVES = 8;
constexpr int GEMB_THREADS = GEMB_WAVES * 64;

struct GembArgs {
  const u32x4_t* w0;
  const u32x4_t* w1;
  const uint32_t* sz0;
  const uint32_t* sz1;
  const void* x;  // FT [M, ldx]
  int ldx;
  const void* bias;
  const void* residual;
  void* y;
  int ldy;
  const float* h_res;
  float* h_out;
  float alpha;
  int act;
  int M, N, K;
  int KT, NTILES, Gp;
  int ktpg;     // k-tiles per quantisation group (per-channel: >= KT)
  int kgroups;  // K-split units (groups or k-tiles)
};

// MT: 16-row tiles (1: M <= 16, 2: M <= 32); PH: k-tiles whose A fragments are resident at a time
template <int WBITS, int FT, int MT, int EPI, int GPT>
__global__ __launch_bounds__(GEMB_THREADS) void gemv_batch_kernel(const GembArgs a) {
  using WT = WTraits<WBITS>;
  using EX = ExpandV<WBITS, FT>;
  constexpr int KSTEPS = WT::KSTEPS;
  constexpr int KTILE = WT::KTILE;
  constexpr bool QUANT = WBITS != 16;
  constexpr int DUAL = EPI == EPI_SWIGLU ? 2 : 1;
  // k-tiles whose A fragments are register-resident at a time: PH * KSTEPS * MT * 4 <= 64 VGPRs.  The
  // kernel must NOT spill: a spill of a register with an asm load in flight saves garbage.
  constexpr int PH = 16 / (KSTEPS * MT) < 8 ? 16 / (KSTEPS * MT) : 8;
  static_assert(PH >= 1 && PH * KSTEPS * MT * 4 <= 64, "A fragment budget");

  __shared__ __attribute__((aligned(16))) float red[DUAL * GEMB_WAVES * 16 * MT * 16];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int ni = lane & 15, kb = lane >> 4;
  const int tile = blockIdx.x;

  // K split in whole quantisation groups
  const bool subc = QUANT && a.ktpg < a.KT;
  const int gsz = subc ? a.ktpg : 1;
  const int gcount = subc ? a.ktpg : (1 << 30);
  const int g_lo = (a.kgroups * wave) / GEMB_WAVES;
  const int k_lo = min(a.KT, g_lo * gsz);
  const int k_hi = min(a.KT, ((a.kgroups * (wave + 1)) / GEMB_WAVES) * gsz);
  const int nk = k_hi - k_lo;

  // ---- per-wave weight / parameter bases ---------------------------------------------------------
  const u32x4_t* wbase[DUAL];
  const uint32_t* sbase[DUAL];
#pragma unroll
  for (int v = 0; v < DUAL; ++v) {
    wbase[v] = (v ? a.w1 : a.w0) + ((size_t)tile * a.KT + k_lo) * 64 + lane;
    sbase[v] = QUANT ? (v ? a.sz1 : a.sz0) + ((size_t)tile * a.Gp + (subc ? g_lo : 0)) * 16 + ni : nullptr;
  }
  // Plain (compiler-visible) loads: the hand-counted asm ring of the batch-1 kernel does not survive
  // the phase structure (register moves of in-flight destinations).  Two register sets alternate:
  // while phase p is multiplied, the weights of phase p+1 are already in flight.
  struct PhaseRegs {
    u32x4_t w[DUAL][PH];
    uint32_t s[DUAL][PH];
  };
  auto load_phase = [&](PhaseRegs& r, int p0) {
#pragma unroll
    for (int v = 0; v < DUAL; ++v)
#pragma unroll
      for (int q = 0; q < PH; ++q) {
        const int kt = min(p0 + q, nk - 1);  // clamped: a valid (re-)load instead of a branch
        r.w[v][q] = __builtin_nontemporal_load(wbase[v] + (size_t)kt * 64);
        if constexpr (QUANT) r.s[v][q] = sbase[v][(size_t)(GPT ? kt : (subc ? kt / gsz : 0)) * 16];
        else r.s[v][q] = 0u;
      }
  };

  // ---- accumulators ------------------------------------------------------------------------------
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t tot[DUAL][MT];
#pragma unroll
  for (int v = 0; v < DUAL; ++v)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) tot[v][mt] = zero4;
  uint32_t ex_mask = 0x000F000Fu, ex_magic = FT == DIHIP_BF16 ? 0x43004300u : 0x64006400u;
  asm volatile("" : "+v"(ex_mask), "+v"(ex_magic));
  const u32x4_t ones = FT == DIHIP_BF16 ? u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}
                                        : u32x4_t{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
  // this lane's activation rows (rows beyond M are clamped: their results are never stored)
  const uint16_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    xrow[mt] = reinterpret_cast<const uint16_t*>(a.x) + (size_t)min(mt * 16 + ni, a.M - 1) * a.ldx + kb * 8 + (size_t)k_lo * KTILE;

  int cgl[DUAL];
  f32x4_t gacc[DUAL][MT], xacc[DUAL][MT];  // running group sums (only when a group spans several k-tiles)
#pragma unroll
  for (int v = 0; v < DUAL; ++v) {
    cgl[v] = gcount;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      gacc[v][mt] = zero4;
      xacc[v][mt] = zero4;
    }
  }

  auto run_phase = [&](const PhaseRegs& r, int p0) {
    const int plen = min(PH, nk - p0);
    // A fragments of the phase straight from L2 + per-k-tile row sums (one MFMA against ones per k-step)
    u32x4_t af[PH][KSTEPS][MT];
    f32x4_t xs[PH][MT];
#pragma unroll
    for (int q = 0; q < PH; ++q) {
      const int kt = min(p0 + q, nk - 1);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          af[q][ks][mt] = *reinterpret_cast<const u32x4_t*>(xrow[mt] + (size_t)kt * KTILE + ks * 32);
    }
    if constexpr (QUANT) {
#pragma unroll
      for (int q = 0; q < PH; ++q)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4_t sx = zero4;
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) sx = mfma16<FT>(af[q][ks][mt], ones, sx);
          xs[q][mt] = sx;
        }
    }
#pragma unroll
    for (int v = 0; v < DUAL; ++v) {
#pragma unroll
      for (int q = 0; q < PH; ++q) {
        if (q < plen) {  // wave-uniform
          f32x4_t g[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) g[mt] = zero4;
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            const u32x4_t bf = EX::frag(r.w[v][q], ks, ex_mask, ex_magic);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) g[mt] = mfma16<FT>(af[q][ks][mt], bf, g[mt]);
          }
          if constexpr (!QUANT) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) tot[v][mt][rr] += g[mt][rr];
          } else {
            const uint32_t szv = r.s[v][q];
            const float s_ = ft_bits_to_f32<FT>(szv & 0xFFFFu);
            const float nzp_ = -(ft_bits_to_f32<FT>(szv >> 16) + EX::OFFSET);
            if constexpr (GPT) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) tot[v][mt][rr] = fmaf(s_, fmaf(nzp_, xs[q][mt][rr], g[mt][rr]), tot[v][mt][rr]);
            } else {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  gacc[v][mt][rr] += g[mt][rr];
                  xacc[v][mt][rr] += xs[q][mt][rr];
                }
              const bool last = p0 + q + 1 == nk;
              if (--cgl[v] == 0 || last) {  // group end (wave-uniform)
                cgl[v] = gcount;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                  for (int rr = 0; rr < 4; ++rr) {
                    tot[v][mt][rr] = fmaf(s_, fmaf(nzp_, xacc[v][mt][rr], gacc[v][mt][rr]), tot[v][mt][rr]);
                    gacc[v][mt][rr] = 0.f;
                    xacc[v][mt][rr] = 0.f;
                  }
              }
            }
          }
        }
      }
    }
  };

  if (nk > 0) {
    PhaseRegs ra, rb;
    load_phase(ra, 0);
    for (int p0 = 0; p0 < nk; p0 += 2 * PH) {
      if (p0 + PH < nk) load_phase(rb, p0 + PH);
      run_phase(ra, p0);
      if (p0 + PH < nk) {
        if (p0 + 2 * PH < nk) load_phase(ra, p0 + 2 * PH);
        run_phase(rb, p0 + PH);
      }
    }
  }

  // ---- combine the 8 k-slices through LDS, fixed order ----
#pragma unroll
  for (int v = 0; v < DUAL; ++v)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((v * GEMB_WAVES + wave) * (16 * MT) + mt * 16 + kb * 4 + r) * 16 + ni] = tot[v][mt][r];
  __syncthreads();
  for (int e = tid; e < a.M * 16; e += GEMB_THREADS) {
    const int col = e & 15, m = e >> 4;
    const int n = tile * 16 + col;
    if (n >= a.N) continue;
    float v = 0.f, v2 = 0.f;
    for (int s = 0; s < GEMB_WAVES; ++s) v += red[(s * (16 * MT) + m) * 16 + col];
    if constexpr (DUAL == 2)
      for (int s = 0; s < GEMB_WAVES; ++s) v2 += red[((GEMB_WAVES + s) * (16 * MT) + m) * 16 + col];
    if constexpr (EPI == EPI_STD) {
      v = __fmul_rn(a.alpha, v);
      if (a.bias) v = __fadd_rn(v, load_ft<FT>(a.bias, n));
      v = apply_act(v, a.act);
      if (a.residual) v = ft_round<FT>(v) + load_ft<FT>(a.residual, (size_t)m * a.ldy + n);
      store_ft<FT>(a.y, (size_t)m * a.ldy + n, v);
    } else if constexpr (EPI == EPI_SWIGLU) {
      store_ft<FT>(a.y, (size_t)m * a.ldy + n, (v / (1.f + expf(-v))) * v2);
    } else {
      const float base = a.h_res ? a.h_res[(size_t)m * a.N + n] : 0.f;
      a.h_out[(size_t)m * a.N + n] = __fadd_rn(base, __fmul_rn(a.alpha, v));
    }
  }
}

template <int WBITS, int FT, int MT, int EPI, int GPT>
hipError_t launch_gemv_batch(const GembArgs& a, int blocks, hipStream_t stream);

#define DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, MT, EPI, GPT)                                        \
  template <>                                                                                    \
  hipError_t launch_gemv_batch<WBITS, FT, MT, EPI, GPT>(const GembArgs& a, int blocks, hipStream_t s) { \
    hipLaunchKernelGGL((gemv_batch_kernel<WBITS, FT, MT, EPI, GPT>), dim3(blocks), dim3(GEMB_THREADS), 0, s, a); \
    return hipGetLastError();                                                                    \
  }
#define DIHIP_DEFINE_GEMB_LAUNCH_SET(WBITS, FT, GPT)        \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 1, EPI_STD, GPT)      \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 2, EPI_STD, GPT)      \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 1, EPI_SWIGLU, GPT)   \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 2, EPI_SWIGLU, GPT)   \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 1, EPI_ADDTO, GPT)    \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 2, EPI_ADDTO, GPT)

}  // namespace dihip
