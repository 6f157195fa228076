// gemm_panel_kernel.hpp -- weight-only GEMM for batched decode (4 < M <= 32) with the activations in the
// FRAG32 layout: a workgroup owns a PANEL of 8 column tiles (one per wave; SwiGLU: gate/up tile pairs) and a
// K-slice, and shares every activation k-tile through LDS.
//
// gemv_batch_kernel.hpp keeps whole columns in one workgroup (no inter-workgroup reduction): its 8 waves
// split K, so every workgroup re-reads the full M x K activations -- 163 MB (gate/up) / 271 MB (down) of
// L2 -> CU traffic per launch at batch 32 next to 72 / 36 MB of weights -- and, because loads return in
// order, the weight stream of a wave cannot run further ahead than its (8x larger) activation stream.
// Here the roles are swapped:
//   * the 8 waves split N; an activation k-tile (FRAG32: KSTEPS*MT contiguous 1 KiB fragments) is copied
//     into LDS ONCE per workgroup, one 16-byte load per lane per wave, and every wave reads its A
//     fragments from there (ds_read_b128, lane-linear, conflict free).  L2 -> CU activation traffic drops
//     8x and a wave's global-load queue holds 1 KiB of x per (1 or 2) KiB of weights;
//   * both streams are prefetched the same P k-tiles ahead in a register ring (plain unconditional loads in
//     a fully unrolled body: hipcc counts them exactly), so waiting for step t leaves P steps in flight;
//   * one workgroup barrier per k-tile (two LDS slots alternate);
//   * no cross-wave reduction: a wave accumulates its tile over the whole K-slice in registers;
//   * layers with few columns (N = 3584: 28 panels) split K across workgroups: f32 partial tiles go to a
//     slab and gemm_panel_reduce_kernel applies the epilogue (fixed summation order, deterministic).
// The arithmetic per element is the one of the other decode kernels (exact integer MFMA, scale / zero-point
// per quantisation group on the f32 accumulator); only the f32 summation order over K differs.
#pragma once
#include <type_traits>

#include "gemv_stream_kernel.hpp"

namespace dihip {

typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int PANEL_WAVES = 8;
constexpr int PANEL_THREADS = PANEL_WAVES * 64;
// prefetch distance in k-tiles: 6 (one weight chunk per step) / 4 (SwiGLU: two chunks per step); 8 measured slower

struct PanelArgs {
  const u32x4_t* w0;
  const u32x4_t* w1;
  const uint32_t* sz0;
  const uint32_t* sz1;
  const void* x;  // FT, FRAG32 [K/32][MT][64][8]
  const void* bias;
  const void* residual;  // row-major [M, N]
  void* y;
  int ldy;
  const float* h_res;
  float* h_out;
  float alpha;
  int act;
  int M, N, K;
  int KT, NTILES, Gp;
  int ktpg;     // k-tiles per quantisation group (per-channel: >= KT)
  int ktps;     // k-tiles per K-slice (multiple of ktpg when groups span several k-tiles)
  int nslices;  // gridDim.y; > 1: partial tiles go to `slab`
  float* slab;  // [nslices][DUAL][M][N] f32
  int yfrag;    // y (EPI_STD / EPI_SWIGLU) in FRAG32
  int nunits;   // gemm_kslice_kernel.hpp: column tiles (SwiGLU: tile pairs) of the whole launch
  // EPI_ADDTO with a split-K slab only: RMSNorm of the finished rows rides on the reduction (gemm_reduce_addto_norm_kernel)
  const void* n_gamma;  // FT [N]; null: plain reduction
  float n_eps;
  void* n_out;          // FT, row-major [M, N] or FRAG32 (n_frag_mt = 1 / 2)
  int n_frag_mt;
  // deferred RMSNorm (gemv_batch_kernel.hpp, GembArgs): producer side -- n_prenorm: n_out = FT(gamma * h_out) WITHOUT 1/rms and
  // n_rowsq[part][32] partial sums of h_out^2 (split-K reduction: part = row block); consumer side (EPI_STD / EPI_SWIGLU) --
  // rowsq != null: the accumulators of row m are multiplied by 1 / sqrt(Sum_p rowsq[p][m] / K + rowsq_eps)
  int n_prenorm;
  float* n_rowsq;
  const float* rowsq;
  int rowsq_parts;
  float rowsq_eps;
  unsigned long long* trace;  // diagnostics (dihip_debug_set_trace; K-slice kernel): [workgroup][8 waves][8] wall-clock stamps, or null
};

template <int FT, int EPI>
__device__ __forceinline__ void panel_epilogue(const PanelArgs& a, int m, int n, float v, float v2, int mt_tiles) {
  if constexpr (EPI == EPI_STD) {
    v = __fmul_rn(a.alpha, v);
    if (a.bias) v = __fadd_rn(v, load_ft<FT>(a.bias, n));
    v = apply_act(v, a.act);
    if (a.residual) v = ft_round<FT>(v) + load_ft<FT>(a.residual, (size_t)m * a.ldy + n);
    store_ft<FT>(a.y, a.yfrag ? act_frag_index(m, n, mt_tiles) : (size_t)m * a.ldy + n, v);
  } else if constexpr (EPI == EPI_SWIGLU) {
    store_ft<FT>(a.y, a.yfrag ? act_frag_index(m, n, mt_tiles) : (size_t)m * a.ldy + n, (v / (1.f + expf(-v))) * v2);
  } else {
    const float base = a.h_res ? a.h_res[(size_t)m * a.N + n] : 0.f;
    a.h_out[(size_t)m * a.N + n] = __fadd_rn(base, __fmul_rn(a.alpha, v));
  }
}

template <int WBITS, int FT, int MT, int EPI, int GPT>
__global__ __launch_bounds__(PANEL_THREADS) void gemm_panel_kernel(const PanelArgs a) {
  using WT = WTraits<WBITS>;
  using EX = ExpandV<WBITS, FT>;
  constexpr int KSTEPS = WT::KSTEPS;
  constexpr int DUAL = EPI == EPI_SWIGLU ? 2 : 1;
  constexpr int NF = KSTEPS * MT;  // activation fragments (1 KiB each) per k-tile: 2, 4 or 8
  constexpr int P = DUAL == 2 ? 4 : 6;
  static_assert(NF <= PANEL_WAVES, "one fragment per wave");

  __shared__ __attribute__((aligned(16))) u32x4_t xlds[2][NF][64];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int ni = lane & 15, kb = lane >> 4;
  const int unit = blockIdx.x * PANEL_WAVES + wave;
  const int tile = min(unit, a.NTILES - 1);  // clamped: a panel's spare waves redo the last tile, never store
  const int kt0 = blockIdx.y * a.ktps;
  const int nk = min(a.KT, kt0 + a.ktps) - kt0;  // the same for all waves: they walk the slice in lock-step

  const bool subc = a.ktpg < a.KT;
  const int gcount = subc ? a.ktpg : (1 << 30);

  // ---- streams: this wave's weight chunks + its share (fragment wave % NF) of the activation k-tile ----------
  struct Slot {
    u32x4_t w[DUAL];
    uint32_t s[DUAL];
    u32x4_t x;
  };
  const u32x4_t* wp[DUAL];
  const uint32_t* sp[DUAL];
#pragma unroll
  for (int v = 0; v < DUAL; ++v) {
    wp[v] = (v ? a.w1 : a.w0) + ((size_t)tile * a.KT + kt0) * 64 + lane;
    sp[v] = (v ? a.sz1 : a.sz0) + (size_t)tile * a.Gp * 16 + ni;
  }
  const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)((size_t)a.KT * NF * 1024), 0x00020000);
  const uint32_t xoff = (uint32_t)(((size_t)kt0 * NF + (wave % NF)) * 64 + lane) * 16u;
  auto load_slot = [&](Slot& r, int t) {
    const int tt = min(t, nk - 1);  // past the slice: a valid re-load (unconditional loads keep hipcc's vmcnt counting exact)
#pragma unroll
    for (int v = 0; v < DUAL; ++v) {
      r.w[v] = __builtin_nontemporal_load(wp[v] + (size_t)tt * 64);
      const int kt = kt0 + tt;
      // (nontemporal builtin: a plain load of read-only memory is rematerialisable -- hipcc then re-loads the word
      // right before its use instead of keeping it in the ring, and waits vmcnt(0) for it)
      r.s[v] = __builtin_nontemporal_load(sp[v] + (size_t)(GPT ? kt : (subc ? kt / a.ktpg : 0)) * 16);
    }
    // the activation k-tile is read by every panel (and K-slice) of the launch: a TEMPORAL load (round 4: L2 hits of nontemporal
    // requests are served slower, gemm_kslice_kernel.hpp), as a buffer load so that hipcc neither rematerialises it nor loses count of it
    r.x = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xoff + (uint32_t)tt * (NF * 1024u), 0, 0));
  };

  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t tot[DUAL][MT], gacc[DUAL][MT], xacc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    xacc[mt] = zero4;
#pragma unroll
    for (int v = 0; v < DUAL; ++v) {
      tot[v][mt] = zero4;
      gacc[v][mt] = zero4;
    }
  }
  int cgl = gcount;
  uint32_t ex_mask = 0x000F000Fu, ex_magic = FT == DIHIP_BF16 ? 0x43004300u : 0x64006400u;
  asm volatile("" : "+v"(ex_mask), "+v"(ex_magic));
  const u32x4_t ones = FT == DIHIP_BF16 ? u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}
                                        : u32x4_t{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};

  Slot ring[P];
#pragma unroll
  for (int j = 0; j < P; ++j) load_slot(ring[j], j);

  // one k-tile step of the workgroup (ring slot j); REFILL: request k-tile t + P into the freed slot
  auto step = [&](Slot& slot, int t, auto refill) {
    constexpr bool REFILL = decltype(refill)::value;
    // publish this wave's share of activation k-tile t.  The ring slot is refilled (k-tile t + P) only AFTER its
    // last use below: a refill before the use makes hipcc rename the registers and rotate the whole ring with
    // v_mov at the loop head -- which has to wait for every load in flight.
    xlds[t & 1][wave % NF][lane] = slot.x;
    __syncthreads();  // slot t & 1 complete; the other slot (k-tile t - 1) is free for the next step's writes
    u32x4_t af[KSTEPS][MT];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[ks][mt] = xlds[t & 1][ks * MT + mt][lane];
    f32x4_t xs[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {  // Sum_k x[m][k] of the k-tile: one MFMA against ones per k-step
      f32x4_t sx = zero4;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) sx = mfma16<FT>(af[ks][mt], ones, sx);
      xs[mt] = sx;
      if constexpr (!GPT)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) xacc[mt][rr] += sx[rr];
    }
    bool gend = t + 1 == nk;
    if constexpr (!GPT) gend = --cgl == 0 || gend;
#pragma unroll
    for (int v = 0; v < DUAL; ++v) {
      f32x4_t g[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) g[mt] = zero4;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const u32x4_t bf = EX::frag(slot.w[v], ks, ex_mask, ex_magic);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) g[mt] = mfma16<FT>(af[ks][mt], bf, g[mt]);
      }
      const float s_ = ft_bits_to_f32<FT>(slot.s[v] & 0xFFFFu);
      const float nzp_ = -(ft_bits_to_f32<FT>(slot.s[v] >> 16) + EX::OFFSET);
      if constexpr (GPT) {
        // explicit {v, v} pairs for the packed FMAs: hipcc otherwise broadcasts ONE register with op_sel and names a
        // 64-bit pair whose other half is whatever sits next to it -- here a ring register with a load in flight,
        // i.e. a false dependency that drains the prefetch queue once per ring revolution
        f32x2_t s2 = {s_, s_}, nz2 = {nzp_, nzp_};
        asm volatile("" : "+v"(s2), "+v"(nz2));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x2_t x0 = {xs[mt][0], xs[mt][1]}, x1 = {xs[mt][2], xs[mt][3]};
          const f32x2_t g0 = {g[mt][0], g[mt][1]}, g1 = {g[mt][2], g[mt][3]};
          const f32x2_t t0v = {tot[v][mt][0], tot[v][mt][1]}, t1v = {tot[v][mt][2], tot[v][mt][3]};
          const f32x2_t r0 = __builtin_elementwise_fma(s2, __builtin_elementwise_fma(nz2, x0, g0), t0v);
          const f32x2_t r1 = __builtin_elementwise_fma(s2, __builtin_elementwise_fma(nz2, x1, g1), t1v);
          tot[v][mt] = f32x4_t{r0[0], r0[1], r1[0], r1[1]};
        }
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) gacc[v][mt][rr] += g[mt][rr];
        if (gend) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              tot[v][mt][rr] = fmaf(s_, fmaf(nzp_, xacc[mt][rr], gacc[v][mt][rr]), tot[v][mt][rr]);
              gacc[v][mt][rr] = 0.f;
            }
        }
      }
    }
    if constexpr (!GPT) {
      if (gend) {
        cgl = gcount;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xacc[mt] = zero4;
      }
    }
    if constexpr (REFILL) {
      // pin the refill between the slot's last use and the next step: hoisted above the use it is renamed and
      // copied back at the loop edge (a copy waits for the load), sunk below it shortens the prefetch distance
      __builtin_amdgcn_sched_barrier(0);
      load_slot(slot, t + P);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // full ring revolutions: straight-line, unconditional loads (a load behind a branch makes hipcc drain the
  // queue at the join); the last nk % P steps run guarded and request nothing
  int t0 = 0;
  for (; t0 + P <= nk; t0 += P) {
#pragma unroll
    for (int j = 0; j < P; ++j) step(ring[j], t0 + j, std::true_type{});
  }
#pragma unroll
  for (int j = 0; j < P; ++j)
    if (t0 + j < nk) step(ring[j], t0 + j, std::false_type{});  // workgroup-uniform

  // ---- rows kb*4 + r, column ni of this wave's tile ----------------------------------------------------------
  if (unit >= a.NTILES) return;
  const int n = tile * 16 + ni;
  if (n >= a.N) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int m = mt * 16 + kb * 4 + rr;
      if (m >= a.M) continue;
      if (a.nslices > 1) {
#pragma unroll
        for (int v = 0; v < DUAL; ++v)
          a.slab[(((size_t)blockIdx.y * DUAL + v) * a.M + m) * a.N + n] = tot[v][mt][rr];
      } else {
        panel_epilogue<FT, EPI>(a, m, n, tot[0][mt][rr], DUAL == 2 ? tot[DUAL - 1][mt][rr] : 0.f, MT);
      }
    }
}

// split-K: sum the slices in fixed order and apply the epilogue
template <int FT, int EPI>
__global__ __launch_bounds__(256) void gemm_panel_reduce_kernel(const PanelArgs a, int mt_tiles) {
  constexpr int DUAL = EPI == EPI_SWIGLU ? 2 : 1;
  const size_t total = (size_t)a.M * a.N;
  // deferred RMSNorm, consumer side: 1 / rms per row from the producer's partial sums.  A workgroup that makes one pass over 256
  // consecutive elements touches at most two rows (N >= 256): 128 threads per row add the parts (thread i: parts i, i + 128,
  // ...), wave sums in DPP order, the two waves of a row in order -- one short round of loads.  Otherwise: all 32 rows, thread
  // (row, part) with 8 parts in LDS.  Every workgroup adds a row's parts in the same order.
  __shared__ float rsp[EPI != EPI_ADDTO ? 8 : 1][32];
  __shared__ float rstd_l[EPI != EPI_ADDTO ? 32 : 1];
  bool rs_on = false;
  if constexpr (EPI != EPI_ADDTO) {
    rs_on = a.rowsq != nullptr;
    if (rs_on) {
      const bool one_pass = (size_t)gridDim.x * 256 >= total && a.N >= 256 && a.rowsq_parts <= 256;
      if (one_pass) {
        const size_t e0 = (size_t)blockIdx.x * 256;
        const int m_lo = (int)(e0 / a.N);
        const int row = min(m_lo + (int)(threadIdx.x >> 7), 31), i = threadIdx.x & 127;
        const float p0 = i < a.rowsq_parts ? a.rowsq[(size_t)i * 32 + row] : 0.f;
        const float p1 = i + 128 < a.rowsq_parts ? a.rowsq[(size_t)(i + 128) * 32 + row] : 0.f;
        const float t = wave_sum(p0 + p1);
        if ((threadIdx.x & 63) == 0) rsp[0][threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x < 2) rstd_l[min(m_lo + (int)threadIdx.x, 31)] = 1.f / sqrtf((rsp[0][2 * threadIdx.x] + rsp[0][2 * threadIdx.x + 1]) / (float)a.K + a.rowsq_eps);
        __syncthreads();
      } else {
        const int m = threadIdx.x & 31, part = threadIdx.x >> 5;
        float t = 0.f;
        for (int p0 = part; p0 < a.rowsq_parts; p0 += 8 * 32) {
          float r[32];
#pragma unroll
          for (int j = 0; j < 32; ++j) r[j] = p0 + j * 8 < a.rowsq_parts ? a.rowsq[(size_t)(p0 + j * 8) * 32 + m] : 0.f;
#pragma unroll
          for (int j = 0; j < 32; ++j) t += r[j];
        }
        rsp[part][m] = t;
        __syncthreads();
        if (threadIdx.x < 32) {
          float tt = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) tt += rsp[q][threadIdx.x];
          rstd_l[threadIdx.x] = 1.f / sqrtf(tt / (float)a.K + a.rowsq_eps);
        }
        __syncthreads();
      }
    }
  }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int m = (int)(e / a.N), n = (int)(e - (size_t)m * a.N);
    float v = 0.f, v2 = 0.f;
    for (int s = 0; s < a.nslices; ++s) {
      v += a.slab[((size_t)s * DUAL) * total + e];
      if constexpr (DUAL == 2) v2 += a.slab[((size_t)s * DUAL + 1) * total + e];
    }
    if constexpr (EPI != EPI_ADDTO) {
      if (rs_on) {
        const float r_ = rstd_l[m];
        v *= r_;
        v2 *= r_;
      }
    }
    panel_epilogue<FT, EPI>(a, m, n, v, v2, mt_tiles);
  }
}

// split-K reduction of an EPI_ADDTO GEMM fused with the RMSNorm that follows it in the decoder graph: one workgroup per
// row keeps the finished f32 row in registers (slices summed in the fixed order of gemm_panel_reduce_kernel, residual
// added as in panel_epilogue), writes h_out, and normalises straight into the next GEMM's activation layout -- the
// arithmetic of rmsnorm_f32_to_ft_kernel (gemm_lowp.hip), one launch and one pass over the row less.
// N % 4 == 0, N <= 4 * THREADS * VPT.  Rows wider than 4096 take 1024 threads (a 256-thread block walked a row of 8192 in eight
// dependent steps: 9.0 us at 16 rows, profiles/r03z cfg3_rank).
template <int FT, int VPT, int THREADS>
__global__ __launch_bounds__(THREADS) void gemm_reduce_addto_norm_kernel(const PanelArgs a) {
  __shared__ float red[THREADS / 64];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int nvec = a.N >> 2;
  const size_t total = (size_t)a.M * a.N;
  f32x4_t v[VPT];
  u32x2_t gm[VPT];
  // loads of SB slices x VPT column vectors are in flight together (a loop over the runtime slice count with one
  // dependent accumulation per load would serialise ~40 memory round trips per thread -- the slab comes from other XCDs'
  // L2s, i.e. from memory); slices are still added in order
  f32x4_t base[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = min(tid + i * THREADS, nvec - 1);  // clamped: lanes past the row re-load its last vector and store nothing
    const size_t e = (size_t)row * a.N + (size_t)c * 4;
    v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    gm[i] = reinterpret_cast<const u32x2_t*>(a.n_gamma)[c];
    base[i] = a.h_res ? *reinterpret_cast<const f32x4_t*>(a.h_res + e) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int SB = VPT <= 4 ? 8 : 4;
  for (int s0 = 0; s0 < a.nslices; s0 += SB) {
    f32x4_t t[SB][VPT];
#pragma unroll
    for (int j = 0; j < SB; ++j) {
      const int sj = min(s0 + j, a.nslices - 1);
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
        const int c = min(tid + i * THREADS, nvec - 1);
        t[j][i] = *reinterpret_cast<const f32x4_t*>(a.slab + (size_t)sj * total + (size_t)row * a.N + (size_t)c * 4);
      }
    }
#pragma unroll
    for (int j = 0; j < SB; ++j)
      if (s0 + j < a.nslices) {
#pragma unroll
        for (int i = 0; i < VPT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i][r] += t[j][i][r];
      }
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = tid + i * THREADS;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[i][r] = c < nvec ? __fadd_rn(base[i][r], __fmul_rn(a.alpha, v[i][r])) : 0.f;
    if (c < nvec) *reinterpret_cast<f32x4_t*>(a.h_out + (size_t)row * a.N + (size_t)c * 4) = v[i];
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) ss += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  float tot = red[0];
#pragma unroll
  for (int w = 1; w < THREADS / 64; ++w) tot += red[w];  // ((r0 + r1) + r2) + ...: the order of rmsnorm_f32_to_ft_kernel
  const float rstd = 1.f / sqrtf(tot / (float)a.N + a.n_eps);
  uint16_t* y = reinterpret_cast<uint16_t*>(a.n_out);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = tid + i * THREADS;
    if (c < nvec) {
      const float g0 = ft_bits_to_f32<FT>(gm[i][0] & 0xFFFFu), g1 = ft_bits_to_f32<FT>(gm[i][0] >> 16);
      const float g2 = ft_bits_to_f32<FT>(gm[i][1] & 0xFFFFu), g3 = ft_bits_to_f32<FT>(gm[i][1] >> 16);
      const uint32_t lo = f32_to_ft_bits<FT>((g0 * v[i][0]) * rstd) | (f32_to_ft_bits<FT>((g1 * v[i][1]) * rstd) << 16);
      const uint32_t hi = f32_to_ft_bits<FT>((g2 * v[i][2]) * rstd) | (f32_to_ft_bits<FT>((g3 * v[i][3]) * rstd) << 16);
      const int k = c * 4;
      const size_t idx = a.n_frag_mt ? act_frag_index(row, k, a.n_frag_mt) : (size_t)row * a.N + k;
      *reinterpret_cast<u32x2_t*>(y + idx) = u32x2_t{lo, hi};
    }
  }
}

// the reduction launch shared by the panel and the K-slice kernels
template <int FT, int EPI>
inline void launch_slab_reduce(const PanelArgs& a, int mt_tiles, hipStream_t s) {
  if constexpr (EPI == EPI_ADDTO) {
    if (a.n_gamma) {  // host contract (run_gemm): N % 4 == 0, N <= 8192, 16-byte aligned rows
      if (a.N <= 4096) hipLaunchKernelGGL((gemm_reduce_addto_norm_kernel<FT, 4, 256>), dim3(a.M), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_reduce_addto_norm_kernel<FT, 2, 1024>), dim3(a.M), dim3(1024), 0, s, a);
      return;
    }
  }
  const int blocks = (int)std::min<size_t>(((size_t)a.M * a.N + 255) / 256, 1024);
  hipLaunchKernelGGL((gemm_panel_reduce_kernel<FT, EPI>), dim3(blocks), dim3(256), 0, s, a, mt_tiles);
}

template <int WBITS, int FT, int MT, int EPI, int GPT>
hipError_t launch_gemm_panel(const PanelArgs& a, int panels, hipStream_t stream);

#define DIHIP_DEFINE_PANEL_LAUNCH(WBITS, FT, MT, EPI, GPT)                                                   \
  template <>                                                                                                \
  hipError_t launch_gemm_panel<WBITS, FT, MT, EPI, GPT>(const PanelArgs& a, int panels, hipStream_t s) {     \
    hipLaunchKernelGGL((gemm_panel_kernel<WBITS, FT, MT, EPI, GPT>), dim3(panels, a.nslices), dim3(PANEL_THREADS), 0, s, a); \
    if (a.nslices > 1) launch_slab_reduce<FT, EPI>(a, MT, s);                                                \
    return hipGetLastError();                                                                                \
  }
#define DIHIP_DEFINE_PANEL_LAUNCH_SET(WBITS, FT, GPT)      \
  DIHIP_DEFINE_PANEL_LAUNCH(WBITS, FT, 1, EPI_STD, GPT)    \
  DIHIP_DEFINE_PANEL_LAUNCH(WBITS, FT, 2, EPI_STD, GPT)    \
  DIHIP_DEFINE_PANEL_LAUNCH(WBITS, FT, 1, EPI_SWIGLU, GPT) \
  DIHIP_DEFINE_PANEL_LAUNCH(WBITS, FT, 2, EPI_SWIGLU, GPT) \
  DIHIP_DEFINE_PANEL_LAUNCH(WBITS, FT, 1, EPI_ADDTO, GPT)  \
  DIHIP_DEFINE_PANEL_LAUNCH(WBITS, FT, 2, EPI_ADDTO, GPT)

}  // namespace dihip
