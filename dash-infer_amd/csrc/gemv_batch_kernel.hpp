// gemv_batch_kernel.hpp -- weight-only GEMM for small decode batches (4 < M <= 32) on gfx950: whole columns per
// workgroup (row-major or FRAG32 activations; the panel kernel, gemm_panel_kernel.hpp, takes the large FRAG32 shapes).
//
// Batched decode (BASELINE configs #3 / #4: M = 32 / 16 rows) streams the same weight bytes as
// batch 1, so it is just as launch-latency sensitive -- but M x K activations (up to 1.2 MB) no
// longer fit in LDS, which the batch-1 kernel (gemv_stream_kernel.hpp) relies on.  This kernel keeps
// the no-inter-workgroup-communication structure of that kernel and changes where the A operand lives:
//
//   * a workgroup (8 waves) owns a.upb consecutive units (a unit = one 16-column tile; SwiGLU: the gate
//     and the up tile of the same columns) and walks them NT at a time; the 8 waves split K and combine
//     through LDS in fixed order -- no inter-workgroup communication, bit-reproducible;
//   * each wave loads the MFMA A fragments of ITS k-range straight from the activation matrix in
//     L2 (lane (kb, row) <- 16 bytes of x[row][k..k+8]) into registers, one k-tile ahead, and uses
//     them for all NT x DUAL tiles of the group; nothing is staged in LDS;
//   * sum_k x[m][k] of each k-tile (needed by the zero-point term) comes from one extra MFMA per
//     k-step against an all-ones B fragment -- rows land in the C layout the fix-up needs;
//   * weights and A fragments of the next k-tile are in flight while the current one is multiplied
//     (two alternating register sets, plain unconditional loads), and are expanded / scaled exactly as
//     in the batch-1 kernel; every row goes through the same arithmetic whatever the batch.
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
  int upb;      // units (column tiles; SwiGLU: gate/up tile pairs) per workgroup
  int xfrag;    // x is in the FRAG32 activation layout (act_frag_index) instead of row-major [M, ldx]
  int yfrag;    // y (EPI_STD / EPI_SWIGLU) is written in the FRAG32 layout
  // Deferred RMSNorm (round 5; dihip_fused_gemm_addto_prenorm / dihip_prenorm_gemm_rowsq): the 1/rms of a row is a scalar that
  // commutes with the GEMM, so the LayerNormNoBeta between a residual GEMM and the next GEMM needs no launch of its own --
  //   producer (EPI_ADDTO, n_out != null): next to h_out the epilogue writes n_out = FT(gamma * h_out) -- no 1/rms -- in the
  //     consumer's layout, and n_rowsq[blockIdx.x][32] = this workgroup's share of Sum_n h_out[m][n]^2 (fixed order);
  //   consumer (EPI_STD, rowsq != null): every accumulator of row m is multiplied by 1 / sqrt(Sum_p rowsq[p][m] / K + eps)
  //     before alpha / bias -- parts summed in a fixed order, every workgroup the same way.
  // The rounding point moves (the reference rounds (gamma * x) * rstd to FT, here gamma * x is rounded and rstd applied to the
  // f32 accumulator): same relative precision, not the same bits -- parity is asserted against the oracle's tolerance.
  const void* n_gamma;
  void* n_out;
  int n_frag_mt;
  float* n_rowsq;
  const float* rowsq;
  int rowsq_parts;
  float rowsq_eps;
};
constexpr int GEMB_MAXU = 8;  // units per workgroup the producer form keeps row partials for



// MT: 16-row tiles (1: M <= 16, 2: M <= 32); NT: column tiles (SwiGLU: gate/up tile pairs) that share
// one pass over the activations.  A workgroup owns a.upb consecutive units and walks them NT at a time.
template <int WBITS, int FT, int MT, int NT, int EPI, int GPT>
__global__ __launch_bounds__(GEMB_THREADS) void gemv_batch_kernel(const GembArgs a) {
  using WT = WTraits<WBITS>;
  using EX = ExpandV<WBITS, FT>;
  constexpr int KSTEPS = WT::KSTEPS;
  constexpr int KTILE = WT::KTILE;
  constexpr bool QUANT = WBITS != 16;
  constexpr int DUAL = EPI == EPI_SWIGLU ? 2 : 1;
  // the cross-wave combine handles CU units at a time (SwiGLU: the gate and up tile of one column tile)
  constexpr int CU = DUAL == 2 ? 2 : NT;
  constexpr int NCHUNK = DUAL == 2 ? NT : 1;

  __shared__ __attribute__((aligned(16))) float red[CU * GEMB_WAVES * 16 * MT * 16];
  __shared__ float rsq[EPI == EPI_ADDTO ? GEMB_MAXU : 1][32];  // producer: Sum h^2 per (unit of this workgroup, row)
  __shared__ float rsp[EPI == EPI_STD ? GEMB_THREADS / 32 : 1][32];  // consumer: partial sums of the producer's parts
  __shared__ float rstd_l[EPI == EPI_STD ? 32 : 1];                  // consumer: 1 / rms per row

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int ni = lane & 15, kb = lane >> 4;
  const int u_lo = blockIdx.x * a.upb;
  const int u_hi = min(a.NTILES, u_lo + a.upb);
  const int ngroups = (u_hi - u_lo + NT - 1) / NT;

  // deferred RMSNorm, consumer side: 1 / rms of every row from the producer's per-workgroup partial sums -- thread (row, part)
  // adds its parts in a fixed order, 16 parts meet in LDS.  The loads go out BEHIND the first k-tile's loads (loads return in
  // order, and the parts come from other XCDs: requested first they would hold the weights back)
  float rs_t[16];
  bool rs_on = false;
  if constexpr (EPI == EPI_STD) rs_on = a.rowsq != nullptr;
  auto rs_request = [&]() {
    if constexpr (EPI == EPI_STD) {
      if (rs_on) {
        const int m = tid & 31, part = tid >> 5;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int pj = part + j * (GEMB_THREADS / 32);
          rs_t[j] = pj < a.rowsq_parts ? a.rowsq[(size_t)pj * 32 + m] : 0.f;
        }
      }
    }
  };

  // K split in whole quantisation groups
  const bool subc = QUANT && a.ktpg < a.KT;
  const int gsz = subc ? a.ktpg : 1;
  const int gcount = subc ? a.ktpg : (1 << 30);
  const int g_lo = (a.kgroups * wave) / GEMB_WAVES;
  const int k_lo = min(a.KT, g_lo * gsz);
  const int k_hi = min(a.KT, ((a.kgroups * (wave + 1)) / GEMB_WAVES) * gsz);
  const int nk = k_hi - k_lo;

  // One k-tile of work: the NT x DUAL weight chunks, their (scale, zero) words and the MFMA A fragments of
  // the activations (lane (kb, row) <- 16 bytes of x[row][k..k+8], straight from L2).  Two sets alternate:
  // while one is multiplied the next k-tile (of this or of the next unit group) is in flight.  The loads are
  // plain, unconditional (the cursor is clamped) and sit in straight-line code, so hipcc counts them exactly
  // (partial s_waitcnt vmcnt(N)); a load behind a branch makes it drain the queue at the join (vmcnt(0)).
  // Loads return in order, so the weight stream cannot run further ahead than the activation stream of the
  // same wave: a deeper weight ring next to one-k-tile-ahead A fragments measured slower.
  struct KtRegs {
    u32x4_t w[NT][DUAL];
    uint32_t s[NT][DUAL];
    u32x4_t af[KSTEPS][MT];
  };
  // this lane's activation rows (rows beyond M are clamped: their results are never stored)
  const uint16_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    xrow[mt] = reinterpret_cast<const uint16_t*>(a.x) + (size_t)min(mt * 16 + ni, a.M - 1) * a.ldx + kb * 8 + (size_t)k_lo * KTILE;

  auto load_kt = [&](KtRegs& r, int grp, int kt) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int tile = min(u_lo + grp * NT + j, a.NTILES - 1);  // clamped: a valid (re-)load instead of a branch
#pragma unroll
      for (int v = 0; v < DUAL; ++v) {
        r.w[j][v] = __builtin_nontemporal_load((v ? a.w1 : a.w0) + ((size_t)tile * a.KT + k_lo + kt) * 64 + lane);
        if constexpr (QUANT) {
          const int gi = GPT ? k_lo + kt : (subc ? g_lo + kt / gsz : 0);
          r.s[j][v] = ((v ? a.sz1 : a.sz0) + ((size_t)tile * a.Gp + gi) * 16)[ni];
        } else {
          r.s[j][v] = 0u;
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const uint16_t* rp = xrow[mt] + (size_t)kt * KTILE + ks * 32;
        const uint16_t* fp = reinterpret_cast<const uint16_t*>(a.x) + ((((size_t)(k_lo + kt) * KSTEPS + ks) * MT + mt) * 64 + lane) * 8;  // act_frag_index
        r.af[ks][mt] = *reinterpret_cast<const u32x4_t*>(a.xfrag ? fp : rp);
      }
  };

  // ---- accumulators ------------------------------------------------------------------------------
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t tot[NT][DUAL][MT];
  f32x4_t gacc[NT][DUAL][MT], xacc[MT];  // running group sums (only when a group spans several k-tiles)
  int cgl = gcount;
  auto reset_acc = [&]() {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int v = 0; v < DUAL; ++v)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          tot[j][v][mt] = zero4;
          gacc[j][v][mt] = zero4;
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xacc[mt] = zero4;
    cgl = gcount;
  };
  reset_acc();
  uint32_t ex_mask = 0x000F000Fu, ex_magic = FT == DIHIP_BF16 ? 0x43004300u : 0x64006400u;
  asm volatile("" : "+v"(ex_mask), "+v"(ex_magic));
  const u32x4_t ones = FT == DIHIP_BF16 ? u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}
                                        : u32x4_t{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};

  // consumer: this thread's parts, added in order, to LDS (once per thread, at its first flush)
  auto rs_stage = [&]() {
    if constexpr (EPI == EPI_STD) {
      if (rs_on) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += rs_t[j];
        rsp[tid >> 5][tid & 31] = t;
      }
    }
  };
  bool rs_first = true;
  // ---- combine the 8 k-slices of a finished unit group through LDS (fixed order) + epilogue ----------
  auto flush = [&](int grp) {
    if (rs_first) rs_stage();  // (the parts have landed long ago; their sums meet behind the first barrier below)
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
#pragma unroll
      for (int u = 0; u < CU; ++u) {
        const int j = DUAL == 2 ? c : u, v = DUAL == 2 ? u : 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((u * GEMB_WAVES + wave) * (16 * MT) + mt * 16 + kb * 4 + r) * 16 + ni] = tot[j][v][mt][r];
      }
      __syncthreads();
      if constexpr (EPI == EPI_STD) {
        if (rs_on && rs_first) {  // (every wave of the workgroup flushes the same number of times: uniform)
          if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < GEMB_THREADS / 32; ++q) t += rsp[q][tid];
            rstd_l[tid] = 1.f / sqrtf(t / (float)a.K + a.rowsq_eps);
          }
          __syncthreads();
        }
        rs_first = false;
      }
      const int per = a.M * 16;
      const int count = DUAL == 2 ? per : per * NT;
      for (int e = tid; e < count; e += GEMB_THREADS) {
        const int u = DUAL == 2 ? 0 : e / per;
        const int rem = e - u * per;
        const int col = rem & 15, m = rem >> 4;
        const int tile = u_lo + grp * NT + (DUAL == 2 ? c : u);
        const int n = tile * 16 + col;
        const bool valid = tile < u_hi && n < a.N;
        bool prep = false;
        if constexpr (EPI == EPI_ADDTO) prep = a.n_out != nullptr;
        if (!valid && !prep) continue;
        float v = 0.f, v2 = 0.f;
        for (int s = 0; s < GEMB_WAVES; ++s) v += red[((u * GEMB_WAVES + s) * (16 * MT) + m) * 16 + col];
        if constexpr (DUAL == 2)
          for (int s = 0; s < GEMB_WAVES; ++s) v2 += red[((GEMB_WAVES + s) * (16 * MT) + m) * 16 + col];
        if constexpr (EPI == EPI_STD) {
          if (rs_on) v *= rstd_l[m];
          v = __fmul_rn(a.alpha, v);
          if (a.bias) v = __fadd_rn(v, load_ft<FT>(a.bias, n));
          v = apply_act(v, a.act);
          if (a.residual) v = ft_round<FT>(v) + load_ft<FT>(a.residual, (size_t)m * a.ldy + n);
          store_ft<FT>(a.y, a.yfrag ? act_frag_index(m, n, MT) : (size_t)m * a.ldy + n, v);
        } else if constexpr (EPI == EPI_SWIGLU) {
          store_ft<FT>(a.y, a.yfrag ? act_frag_index(m, n, MT) : (size_t)m * a.ldy + n, (v / (1.f + expf(-v))) * v2);
        } else {
          float hv = 0.f;
          if (valid) {
            const float base = a.h_res ? a.h_res[(size_t)m * a.N + n] : 0.f;
            hv = __fadd_rn(base, __fmul_rn(a.alpha, v));
            a.h_out[(size_t)m * a.N + n] = hv;
          }
          if (prep) {  // (whole 16-lane rows of one (unit, row) reach this point together: count is a multiple of 16)
            if (valid)
              store_ft<FT>(a.n_out, a.n_frag_mt ? act_frag_index(m, n, a.n_frag_mt) : (size_t)m * a.N + n, load_ft<FT>(a.n_gamma, n) * hv);
            float sq = hv * hv;
            sq += dpp_f32<0xB1>(sq);
            sq += dpp_f32<0x4E>(sq);
            sq += dpp_f32<0x141>(sq);
            sq += dpp_f32<0x140>(sq);
            const int lu = grp * NT + u;
            if (col == 0 && lu < GEMB_MAXU) rsq[lu][m] = sq;
          }
        }
      }
      __syncthreads();
    }
    reset_acc();
  };
  // producer: this workgroup's row partials, units added in order
  auto finish = [&]() {
    if constexpr (EPI == EPI_ADDTO) {
      if (a.n_rowsq && tid < 32) {
        float t = 0.f;
        for (int lu = 0; lu < u_hi - u_lo; ++lu) t += rsq[lu][tid];
        a.n_rowsq[(size_t)blockIdx.x * 32 + tid] = tid < a.M ? t : 0.f;
      }
    }
  };

  auto run_kt = [&](const KtRegs& r, int grp, int kt) {
    f32x4_t xs[MT];
    if constexpr (QUANT) {  // Sum_k x[m][k] of the k-tile: one MFMA against ones per k-step, rows in the C layout
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4_t sx = zero4;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) sx = mfma16<FT>(r.af[ks][mt], ones, sx);
        xs[mt] = sx;
        if constexpr (!GPT)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) xacc[mt][rr] += sx[rr];
      }
    }
    const bool last = kt + 1 == nk;
    bool gend = last;  // quantisation group ends with this k-tile (wave-uniform)
    if constexpr (QUANT && !GPT) gend = --cgl == 0 || last;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int v = 0; v < DUAL; ++v) {
        f32x4_t g[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) g[mt] = zero4;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const u32x4_t bf = EX::frag(r.w[j][v], ks, ex_mask, ex_magic);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) g[mt] = mfma16<FT>(r.af[ks][mt], bf, g[mt]);
        }
        if constexpr (!QUANT) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) tot[j][v][mt][rr] += g[mt][rr];
        } else {
          const uint32_t szv = r.s[j][v];
          const float s_ = ft_bits_to_f32<FT>(szv & 0xFFFFu);
          const float nzp_ = -(ft_bits_to_f32<FT>(szv >> 16) + EX::OFFSET);
          if constexpr (GPT) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) tot[j][v][mt][rr] = fmaf(s_, fmaf(nzp_, xs[mt][rr], g[mt][rr]), tot[j][v][mt][rr]);
          } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) gacc[j][v][mt][rr] += g[mt][rr];
            if (gend) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  tot[j][v][mt][rr] = fmaf(s_, fmaf(nzp_, xacc[mt][rr], gacc[j][v][mt][rr]), tot[j][v][mt][rr]);
                  gacc[j][v][mt][rr] = 0.f;
                }
            }
          }
        }
      }
    if constexpr (QUANT && !GPT) {
      if (gend) {
        cgl = gcount;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xacc[mt] = zero4;
      }
    }
    if (last) flush(grp);
  };

  if (nk == 0) {  // more waves than K groups: this wave only takes part in the combines
    rs_request();
    for (int grp = 0; grp < ngroups; ++grp) flush(grp);
    finish();
    return;
  }
  const int total = ngroups * nk;
  int lg = 0, lk = 0, rg = 0, rk = 0;  // load / run cursors over (unit group, k-tile)
#define DIHIP_GEMB_ADV(G, K) \
  do {                       \
    if (++K == nk) {         \
      K = 0;                 \
      ++G;                   \
    }                        \
  } while (0)
  KtRegs ra, rb;
  load_kt(ra, lg, lk);
  DIHIP_GEMB_ADV(lg, lk);
  rs_request();
  int it = 0;
  for (; it + 1 < total; it += 2) {
    load_kt(rb, min(lg, ngroups - 1), lk);
    DIHIP_GEMB_ADV(lg, lk);
    run_kt(ra, rg, rk);
    DIHIP_GEMB_ADV(rg, rk);
    load_kt(ra, min(lg, ngroups - 1), lk);
    DIHIP_GEMB_ADV(lg, lk);
    run_kt(rb, rg, rk);
    DIHIP_GEMB_ADV(rg, rk);
  }
  if (it < total) run_kt(ra, rg, rk);
#undef DIHIP_GEMB_ADV
  finish();
}

template <int WBITS, int FT, int MT, int NT, int EPI, int GPT>
hipError_t launch_gemv_batch(const GembArgs& a, int blocks, hipStream_t stream);

#define DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, MT, NT, EPI, GPT)                                    \
  template <>                                                                                    \
  hipError_t launch_gemv_batch<WBITS, FT, MT, NT, EPI, GPT>(const GembArgs& a, int blocks, hipStream_t s) { \
    hipLaunchKernelGGL((gemv_batch_kernel<WBITS, FT, MT, NT, EPI, GPT>), dim3(blocks), dim3(GEMB_THREADS), 0, s, a); \
    return hipGetLastError();                                                                    \
  }
#define DIHIP_DEFINE_GEMB_LAUNCH_SET(WBITS, FT, GPT)           \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 1, 1, EPI_STD, GPT)      \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 2, 1, EPI_STD, GPT)      \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 1, 1, EPI_SWIGLU, GPT)   \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 2, 1, EPI_SWIGLU, GPT)   \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 1, 1, EPI_ADDTO, GPT)    \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 2, 1, EPI_ADDTO, GPT)    \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 1, 2, EPI_STD, GPT)      \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 2, 2, EPI_STD, GPT)      \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 1, 2, EPI_SWIGLU, GPT)   \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 2, 2, EPI_SWIGLU, GPT)   \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 1, 2, EPI_ADDTO, GPT)    \
  DIHIP_DEFINE_GEMB_LAUNCH(WBITS, FT, 2, 2, EPI_ADDTO, GPT)

}  // namespace dihip
