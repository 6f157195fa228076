// gemm_kslice_kernel.hpp -- weight-only GEMM for batched decode (4 < M <= 32), activations in the FRAG32 layout,
// with the activations REGISTER-RESIDENT: the waves of a workgroup split K, a wave loads the A fragments of its
// K-slice (16 k-steps: 16 KiB of x at M = 32) once, keeps them for the whole kernel, and streams the weight
// chunks of that slice for every column tile the workgroup owns.
//
// Why a third small-batch kernel: gemm_panel_kernel.hpp shares an activation k-tile through LDS and pays one
// workgroup barrier per k-tile (1-2 KiB of weights per wave), and its weight ring shares the in-order load queue
// with the activation stream; gemv_batch_kernel.hpp re-reads the activations for every column tile.  Here the
// main loop issues nothing but weight loads (the structure of the M <= 4 kernel, which streams at 4.8 TB/s):
//   * a half-unit is the wave's CH chunks of one column tile; half-units come in PAIRS (SwiGLU: the gate and the
//     up tile of one unit, otherwise two tiles).  The ring holds 8 KiB per wave (W4: a pair, W8: one half-unit);
//     the fully unrolled body indexes ring slots and activation registers statically, and every slot is refilled
//     with the same chunk position one ring revolution ahead right after its last use;
//   * the K-slices of a pair meet in LDS (4 KiB per wave, two buffers, ONE barrier per pair = per 8 KiB of weights
//     per wave); 2 * MT waves sum the slices in fixed order and apply the epilogue while the others stream on;
//   * K beyond 8 slices (down projection) is split across workgroups: f32 partial tiles go to the slab of the
//     panel kernel and gemm_panel_reduce_kernel finishes (same layout, same deterministic order).
// Sum_k x[m][k] per chunk (the zero-point term) is computed once per wave with MFMAs against ones and kept in
// wave-private LDS (W4; W8 slices have twice the chunks and recompute it).
// Arithmetic per element as in the other decode kernels (exact integer MFMA, scale / zero-point per group on
// the f32 accumulator); a group that straddles two K-slices is closed in both with the same (scale, zero).
//
// Round 4 (what-if timing above KSL_WAVES' definition):
//   * the activation fragments are loaded WITHOUT the nontemporal hint (-3 us on the 7B gate/up pair at M = 32, -2 at M = 16, with
//     the same asm issue pattern either way: profiles/r04ai_*).  Not because `nt` lines miss: TCC_HIT / TCC_MISS / TCC_REQ of the
//     launch are IDENTICAL with and without the bit (531k hits, 644k misses of 128 B: the 256 re-reads of the fragments hit L2 in
//     both builds, profiles/r04ah_*) -- the hits of nontemporal requests are simply served slower on this part.  The loads are
//     inline asm because a plain builtin load of read-only memory is rematerialised by hipcc right before its use;
//   * tried and removed: an eighth wave WITHOUT a K-slice (K = 3584 has 7) that does the cross-slice sums and the epilogues, so
//     that no streaming wave reaches the next barrier late by a reduction: +3 % stand-alone at M = 16, -5 % at M = 32 (one wave
//     then runs both fragments' SwiGLU epilogues), and in the decode step 11.9k against 12.2k tokens/s (profiles/r04q_*);
//   * tried and removed: the weight ring in LDS filled by LDS-DMA (global_load_lds_dwordx4; 8 KiB per wave in flight at M = 32
//     instead of the 4 KiB the 128 activation registers leave) -- bit-identical, 3 - 7 % SLOWER (profiles/r04n_*): the extra
//     LDS round trip per chunk costs more than the deeper ring gives, because the ring is not what the kernel waits for.
#pragma once
#include "gemm_panel_kernel.hpp"

namespace dihip {

constexpr int KSL_WAVES = 8;
// what-if builds for timing (tools/build_ksl_variant.sh; results are WRONG with any bit set; never in the product build):
//   1 no exchange / barrier / reduction   2 no dequantisation + MFMA (the slot is consumed by an XOR)   4 no activation loads
// Round 4 at M = 32 on the 7B gate/up pair (profiles/r04o_kslice_whatif.txt; launch pair incl. the 4.5 us RMSNorm kernel):
// 29.8 us as built; without (1) 25.0, (2) 27.0, (4) 22.1, all three 16.4 = the weights alone at 6.3 TB/s -- the parts add up,
// i.e. nothing overlaps, and the activation fragments every workgroup pulls are the largest of them.
#ifndef DIHIP_KSL_X
#define DIHIP_KSL_X 0
#endif


// RS = 1: the deferred-RMSNorm consumer (PanelArgs::rowsq; unsplit K, EPI_STD / EPI_SWIGLU).  Its own instantiations: with the code
// merely compiled in, the plain launches ran 1.5 - 2 us slower (main-loop scheduling; profiles/r05_deferred_norm.txt).
template <int WBITS, int FT, int MT, int EPI, int GPT, int RS = 0>
__global__ __launch_bounds__(KSL_WAVES * 64) void gemm_kslice_kernel(const PanelArgs a) {
  using WT = WTraits<WBITS>;
  using EX = ExpandV<WBITS, FT>;
  constexpr int KSTEPS = WT::KSTEPS;
  constexpr int CH = 16 / KSTEPS;  // chunks (k-tiles) per K-slice: 16 k-steps of 32
  constexpr int XS = CH * KSTEPS;
  constexpr int DUAL = EPI == EPI_SWIGLU ? 2 : 1;
  // half-units per ring revolution: a pair at MT = 1 (W4: 8 KiB in flight per wave, W8: 16 KiB); at MT = 2 the 128
  // activation registers leave room for one half-unit (W4: 4 KiB, W8: 8 KiB).  Two pairs (16 KiB, W4) measured 15 %
  // SLOWER than one: the work per workgroup is only ~5 pairs, and whole bodies round it up.
  constexpr int RH = MT == 1 ? 2 : 1;
  constexpr int PU = RH >= 2 ? RH / 2 : 1;  // pairs per unrolled body (static ring slots)
  constexpr int P = RH * CH;

  __shared__ __attribute__((aligned(16))) f32x4_t xch[2][KSL_WAVES][2][MT][64];
  // W4: Sum_k x[m][k] of every chunk of the wave's slice, computed once (wave-private LDS, no barrier): at M = 32 the
  // 8 extra MFMAs per chunk were half of the matrix-pipe time of a kernel that is instruction-bound, not load-bound.
  // W8 slices have 8 chunks (2 x the LDS); their sums are recomputed per chunk.
  constexpr bool XS_LDS = WBITS == 4;
  __shared__ __attribute__((aligned(16))) f32x4_t xsl[XS_LDS ? KSL_WAVES : 1][XS_LDS ? CH : 1][MT][64];
  // deferred RMSNorm, consumer side (PanelArgs::rowsq; unsplit K only -- with a slab the reduction kernel applies it): the
  // producer's partial sums per row are added by thread (row, part) in a fixed order -- 16 loads requested right behind the
  // activation loads and added behind their wait -- and meet in rsp; the epilogue lanes finish 1 / rms for their own rows.
  // (Measured: a slice-less helper wave +1.7 us, adding the parts before the activation loads +1.3, requesting them at entry
  // +0.7 at M = 32 / +2.1 at M = 16 -- loads return in order; profiles/r05_deferred_norm.txt.)
  __shared__ float rsp[RS ? 16 : 1][32];
  __shared__ float rstd_l[RS ? 32 : 1];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int ni = lane & 15, kb = lane >> 4;
  const int nsl_total = a.KT / CH;
  const int slice0 = blockIdx.y * KSL_WAVES;
  const int nw = min(KSL_WAVES, nsl_total - slice0);  // K-slices (= working waves) of this workgroup
  const bool active = wave < nw;
  const int kt0 = min(slice0 + wave, nsl_total - 1) * CH;
  // per-wave wall-clock stamps for tools/kslice_trace.py; compiled in with -DDIHIP_KSL_TRACE only (they cost the
  // MT = 2 variants 4-6 spilled registers)
#ifdef DIHIP_KSL_TRACE
#define DIHIP_KSL_STAMP(I)                                                                                          \
  do {                                                                                                              \
    if (a.trace && lane == 0)                                                                                       \
      a.trace[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * KSL_WAVES + wave) * 8 + (I)] = wall_clock64();       \
  } while (0)
#else
#define DIHIP_KSL_STAMP(I) do { } while (0)
#endif
  DIHIP_KSL_STAMP(0);  // entry
  // (nothing of this is kept live across the main loop: the MT = 2 variants have no register to spare -- the epilogue
  // re-derives what it needs from the kernel arguments)
  static_assert(!RS || EPI != EPI_ADDTO, "deferred row norms: consumer epilogues only");
  auto rs_parts = [&]() { return min((int)blockDim.x >> 5, 16); };
  float rs_r[16];  // the first 16 parts of thread (row, part): requested BEHIND the ring and the activation loads (loads return in order, and the
                   // parts were written by the previous launch on other XCDs: requested first they held everything else back, +0.7 .. 2 us)
  // thread (row, part) adds parts part, part + nparts, ... in that order
  auto rs_sum = [&]() {
    const int rs_nparts = rs_parts();
    if (tid >= rs_nparts * 32) return;
    const int m = tid & 31, part = tid >> 5;
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += rs_r[j];
    for (int p0 = part + 16 * rs_nparts; p0 < a.rowsq_parts; p0 += rs_nparts * 16) {
      float r[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) r[j] = p0 + j * rs_nparts < a.rowsq_parts ? a.rowsq[(size_t)(p0 + j * rs_nparts) * 32 + m] : 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) t += r[j];
    }
    rsp[part][m] = t;
  };
  auto rs_request = [&]() {
    const int rs_nparts = rs_parts();
    const int m = tid & 31, part = min(tid >> 5, rs_nparts - 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) rs_r[j] = part + j * rs_nparts < a.rowsq_parts ? a.rowsq[(size_t)(part + j * rs_nparts) * 32 + m] : 0.f;
  };
  // units of this workgroup: blockIdx.x, + gridDim.x, ...; a unit is a column tile (SwiGLU: the gate / up tile pair)
  const int NB = gridDim.x;
  const int nu = (a.nunits - (int)blockIdx.x + NB - 1) / NB;
  const int NH = nu * DUAL;                                // half-units
  const int NP = (NH + 1) >> 1;                            // pairs
  auto tile_of = [&](int h) {                              // column tile of half-unit h (clamped: dummies re-load the last)
    const int hc = min(h, NH - 1);
    return (int)blockIdx.x + (DUAL == 2 ? hc >> 1 : hc) * NB;
  };

  struct Slot {
    u32x4_t w;
    uint32_t s;
  };
  const bool subc = a.ktpg < a.KT;
  // bases of the half-unit the refills request (wave-uniform tile arithmetic, lane offsets folded in)
  const u32x4_t* wbn;
  const uint32_t* sbn;
  auto set_half = [&](int h) {
    const int tile = tile_of(h);
    const bool up = DUAL == 2 && (min(h, NH - 1) & 1);
    wbn = (up ? a.w1 : a.w0) + ((size_t)tile * a.KT + kt0) * 64 + lane;
    sbn = (up ? a.sz1 : a.sz0) + (size_t)tile * a.Gp * 16 + ni;
  };
  auto load_slot = [&](Slot& r, int c) {  // chunk c of the half-unit set_half() named
    r.w = __builtin_nontemporal_load(wbn + (size_t)c * 64);
    const int kt = kt0 + c;
    r.s = __builtin_nontemporal_load(sbn + (size_t)(GPT ? kt : (subc ? kt / a.ktpg : 0)) * 16);
  };

  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  uint32_t ex_mask = 0x000F000Fu, ex_magic = FT == DIHIP_BF16 ? 0x43004300u : 0x64006400u;
  asm volatile("" : "+v"(ex_mask), "+v"(ex_magic));
  const u32x4_t ones = FT == DIHIP_BF16 ? u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}
                                  : u32x4_t{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};

  // Prologue order: the weight ring is requested FIRST -- its HBM latency then runs under the activation loads, which
  // all workgroups aim at the same 115-229 KB of L2 lines (256 workgroups x 229 KB = 58 MB through L2 at M = 32: the
  // per-wave stamps of tools/kslice_trace.py show ~3 us until the loads are issued and ~5 us until the first MFMA, 20 %
  // of the kernel -- the price of register-resident activations in every workgroup).  The fragments are nontemporal
  // builtin loads: not rematerialisable, so they stay in their registers.
  Slot ring[P];
  if (active) {
#pragma unroll
    for (int r = 0; r < RH; ++r) {
      set_half(r);
#pragma unroll
      for (int c = 0; c < CH; ++c) load_slot(ring[r * CH + c], c);
    }
  }
  // ---- this wave's activations: XS k-steps x MT row tiles, 1 KiB contiguous per fragment ----
  u32x4_t xf[XS][MT];
  if (active) {
    const u32x4_t* xp = reinterpret_cast<const u32x4_t*>(a.x) + (size_t)kt0 * KSTEPS * MT * 64 + lane;
#pragma unroll
    for (int i = 0; i < XS; ++i)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if constexpr (DIHIP_KSL_X & 4) {
          xf[i][mt] = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
          asm volatile("" : "+v"(xf[i][mt]));
        } else {
          // temporal (every workgroup re-reads these lines from L2), not rematerialisable
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xf[i][mt]) : "v"(xp + (size_t)(i * MT + mt) * 64) : "memory");
        }
      }
    if constexpr (RS) {
      rs_request();
    }
    // hipcc does not count asm loads: one explicit wait (the sums below need every fragment, and the ring, requested
    // earlier, has landed by then as well)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < XS; ++i)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(xf[i][mt]));
  }
  if constexpr (RS) {
    if (!active) rs_request();  // (a slice-less wave)
  }
  DIHIP_KSL_STAMP(1);  // ring + activation loads issued
  if constexpr (RS) {  // (the 16 registers are live in the prologue only, where even the MT = 2 variants have room)
    rs_sum();
  }
  if (active) {
    if constexpr (XS_LDS) {
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4_t sx = zero4;
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) sx = mfma16<FT>(xf[c * KSTEPS + ks][mt], ones, sx);
          xsl[wave][c][mt][lane] = sx;
        }
    }
  }

  if constexpr (RS) {
    // one barrier of the prologue: the parts' sums are in rsp; 32 lanes of the LAST wave (never an epilogue wave with >= 3 waves)
    // finish 1 / rms per row -- 16 LDS reads in flight -- and the epilogues read rstd_l behind the first pair's barrier
    __syncthreads();
    if (wave == (int)(blockDim.x >> 6) - 1 && lane < 32) {
      const int np_ = rs_parts();
      float pv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) pv[q] = rsp[min(q, np_ - 1)][lane];
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += q < np_ ? pv[q] : 0.f;
      rstd_l[lane] = 1.f / sqrtf(t / (float)a.K + a.rowsq_eps);
    }
  }
  DIHIP_KSL_STAMP(2);  // activation sums done: ring and activations have landed
  // NP is rounded up to whole bodies: a dummy pair re-loads the last tiles and stores nothing (a guarded pair would put
  // its loads behind a branch, and hipcc drains the queue at such a join)
  for (int p0 = 0; p0 < NP; p0 += PU)
#pragma unroll
  for (int pu = 0; pu < PU; ++pu) {
    const int p = p0 + pu;
    if (active) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        set_half(2 * p + hh + RH);  // the refills below request the half-unit RH ahead (past the end: a valid re-load)
        f32x4_t tot[MT], gacc[MT], xacc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) tot[mt] = gacc[mt] = xacc[mt] = zero4;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          Slot& slot = ring[((2 * pu + hh) % RH) * CH + c];
          f32x4_t g[MT], xs[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) g[mt] = xs[mt] = zero4;
          if constexpr (DIHIP_KSL_X & 2) {
            g[0][0] = __uint_as_float((slot.w[0] ^ slot.w[1] ^ slot.w[2] ^ slot.w[3]) & 0x3FFFFFFFu);
          } else {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
              const u32x4_t bf = EX::frag(slot.w, ks, ex_mask, ex_magic);
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
                g[mt] = mfma16<FT>(xf[c * KSTEPS + ks][mt], bf, g[mt]);
                if constexpr (!XS_LDS) xs[mt] = mfma16<FT>(xf[c * KSTEPS + ks][mt], ones, xs[mt]);  // Sum_k x[m][k] of the chunk
              }
            }
          }
          if constexpr (XS_LDS)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xs[mt] = xsl[wave][c][mt][lane];
          const float s_ = ft_bits_to_f32<FT>(slot.s & 0xFFFFu);
          const float nzp_ = -(ft_bits_to_f32<FT>(slot.s >> 16) + EX::OFFSET);
          if constexpr (GPT) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) tot[mt][rr] = fmaf(s_, fmaf(nzp_, xs[mt][rr], g[mt][rr]), tot[mt][rr]);
          } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) {
                gacc[mt][rr] += g[mt][rr];
                xacc[mt][rr] += xs[mt][rr];
              }
            // the group ends with this chunk, or the slice does (the next slice closes its part with the same scale)
            if (c == CH - 1 || (kt0 + c + 1) % a.ktpg == 0) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                  tot[mt][rr] = fmaf(s_, fmaf(nzp_, xacc[mt][rr], gacc[mt][rr]), tot[mt][rr]);
                  gacc[mt][rr] = 0.f;
                  xacc[mt][rr] = 0.f;
                }
            }
          }
          // refill pinned between the slot's last use and the next chunk (see gemm_panel_kernel.hpp)
          __builtin_amdgcn_sched_barrier(0);
          load_slot(slot, c);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (DIHIP_KSL_X & 1) {
          if (tot[0][0] == 1.2345e-30f) a.slab[lane] = tot[MT - 1][1];  // keeps the sums alive
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) xch[p & 1][wave][hh][mt][lane] = tot[mt];
        }
      }
    }
    if (p == 0) DIHIP_KSL_STAMP(3);  // first pair streamed
    if constexpr (DIHIP_KSL_X & 1) continue;
    __syncthreads();  // the pair's slices are in xch[p & 1]; its previous use (pair p - 2) was reduced before barrier p - 1
    if (p == 0) DIHIP_KSL_STAMP(4);  // first barrier passed
    // ---- reduce over the K-slices in fixed order + epilogue: fragment f = (half, row tile) per wave ----
    constexpr int NFR = DUAL == 2 ? MT : 2 * MT;
    for (int f = wave; f < NFR; f += (int)(blockDim.x >> 6)) {
      const int mt = DUAL == 2 ? f : f % MT;
      const int hh = DUAL == 2 ? 0 : f / MT;
      const int h = 2 * p + hh;
      if (h >= NH) continue;  // the odd last pair's dummy half
      f32x4_t v = zero4, v2 = zero4;
      for (int w = 0; w < nw; ++w) {
        const f32x4_t t = xch[p & 1][w][hh][mt][lane];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) v[rr] += t[rr];
        if constexpr (DUAL == 2) {
          const f32x4_t t2 = xch[p & 1][w][1][mt][lane];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) v2[rr] += t2[rr];
        }
      }
      const int n = tile_of(h) * 16 + ni;
      if (n >= a.N) continue;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int m = mt * 16 + kb * 4 + rr;
        if (m >= a.M) continue;
        if (a.nslices > 1) {
          a.slab[(((size_t)blockIdx.y * DUAL) * a.M + m) * a.N + n] = v[rr];
          if constexpr (DUAL == 2) a.slab[(((size_t)blockIdx.y * DUAL + 1) * a.M + m) * a.N + n] = v2[rr];
        } else {
          if constexpr (RS) {
            int mo = m;
            asm volatile("" : "+v"(mo));  // (opaque: the LDS address is formed here, not hoisted into a register that lives across the main loop)
            const float r_ = rstd_l[mo];
            panel_epilogue<FT, EPI>(a, m, n, v[rr] * r_, v2[rr] * r_, MT);
          } else {
            panel_epilogue<FT, EPI>(a, m, n, v[rr], v2[rr], MT);
          }
        }
      }
    }
  }
  DIHIP_KSL_STAMP(5);  // end
#undef DIHIP_KSL_STAMP
}

template <int WBITS, int FT, int MT, int EPI, int GPT>
hipError_t launch_gemm_kslice(const PanelArgs& a, int groups, int waves, hipStream_t stream);
// the deferred-RMSNorm consumer (bf16, unsplit K, a.rowsq set)
template <int WBITS, int MT, int EPI, int GPT>
hipError_t launch_gemm_kslice_rs(const PanelArgs& a, int groups, int waves, hipStream_t stream);
#define DIHIP_DEFINE_KSLICE_RS_LAUNCH(WBITS, MT, EPI, GPT)                                                     \
  template <>                                                                                                 \
  hipError_t launch_gemm_kslice_rs<WBITS, MT, EPI, GPT>(const PanelArgs& a, int groups, int waves, hipStream_t s) { \
    hipLaunchKernelGGL((gemm_kslice_kernel<WBITS, DIHIP_BF16, MT, EPI, GPT, 1>), dim3(groups, 1), dim3(waves * 64), 0, s, a); \
    return hipGetLastError();                                                                                 \
  }
#define DIHIP_DEFINE_KSLICE_RS_LAUNCH_SET(WBITS, GPT)      \
  DIHIP_DEFINE_KSLICE_RS_LAUNCH(WBITS, 1, EPI_STD, GPT)    \
  DIHIP_DEFINE_KSLICE_RS_LAUNCH(WBITS, 2, EPI_STD, GPT)    \
  DIHIP_DEFINE_KSLICE_RS_LAUNCH(WBITS, 1, EPI_SWIGLU, GPT) \
  DIHIP_DEFINE_KSLICE_RS_LAUNCH(WBITS, 2, EPI_SWIGLU, GPT)

#define DIHIP_DEFINE_KSLICE_LAUNCH(WBITS, FT, MT, EPI, GPT)                                                   \
  template <>                                                                                                 \
  hipError_t launch_gemm_kslice<WBITS, FT, MT, EPI, GPT>(const PanelArgs& a, int groups, int waves, hipStream_t s) { \
    hipLaunchKernelGGL((gemm_kslice_kernel<WBITS, FT, MT, EPI, GPT>), dim3(groups, a.nslices), dim3(waves * 64), 0, s, a); \
    if (a.nslices > 1) launch_slab_reduce<FT, EPI>(a, MT, s);                                                 \
    return hipGetLastError();                                                                                 \
  }
#define DIHIP_DEFINE_KSLICE_LAUNCH_SET(WBITS, FT, GPT)      \
  DIHIP_DEFINE_KSLICE_LAUNCH(WBITS, FT, 1, EPI_STD, GPT)    \
  DIHIP_DEFINE_KSLICE_LAUNCH(WBITS, FT, 2, EPI_STD, GPT)    \
  DIHIP_DEFINE_KSLICE_LAUNCH(WBITS, FT, 1, EPI_SWIGLU, GPT) \
  DIHIP_DEFINE_KSLICE_LAUNCH(WBITS, FT, 2, EPI_SWIGLU, GPT) \
  DIHIP_DEFINE_KSLICE_LAUNCH(WBITS, FT, 1, EPI_ADDTO, GPT)  \
  DIHIP_DEFINE_KSLICE_LAUNCH(WBITS, FT, 2, EPI_ADDTO, GPT)

}  // namespace dihip
