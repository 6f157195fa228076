// decode_attn_block.hip -- the attention half of a batch-1 decode layer in ONE launch (include/dashinfer_hip.h section 3e):
//     RMSNorm + qkv GEMV (+bias)  ->  Rotary + DecoderCacheAppend + paged attention + split merge  ->  o-projection + residual
// i.e. LayerNormNoBeta -> Gemm[A16W4](qkv) -> Rotary -> DecOptMQA -> Gemm[A16W4](o) -> Binary ADD of the reference graph
// (python/pyhie/allspark/model/qwen_v15.py:210-300; the operator loop it shortens: csrc/core/model/model.cpp:1248-1325).
//
// Why: in the launch chain these three operators are 20.2 of the layer's 44.5 us while moving 15 % of its bytes (profiles/r04zzzz_*):
// every launch pays launch -> first byte (~1 us), its own dependent round trips and a tail, and none of the 19 MB they read
// (8.3 MB qkv weights, 4.2 MB of K / V, 6.4 MB o weights) depends on what the previous operator computes.  Here ALL of it is
// requested in the first microsecond of one launch -- it fits the register files: <= 8 KiB per wave -- and what remains
// serial is the arithmetic and three hand-offs:
//
//   workgroups [0, NA)        attention: one (KV group, split) each, 4 waves (the body of span_attn_ft_mfma.hpp, bit for bit):
//                             K / V tiles into registers at entry; q / k / v of this step arrive as GRANULES from the GEMV
//                             workgroups; partial records -> arrival ticket -> the last split of a group merges (as the
//                             stand-alone kernel) and publishes the merged heads as granules + one flag word per group.
//   workgroups [NA, NA + NG)  GEMV: RMSNorm prologue + this workgroup's qkv column tiles from a register-resident weight share
//                             (the K split of the stand-alone decode GEMV, so the sums are bit-identical to it), results
//                             published as granules; its o-projection share has been in registers since entry: it waits for
//                             the group flags, sweeps the attention output granules into LDS and finishes h += attn . Wo.
//
// Granule = 8 bytes {value, tag} written by ONE agent-scope store and read by agent-scope loads: the data is the flag (MI355X
// guide, Guideline 16 form R2), so no fences and nothing outside the memory model.  tag = the launch's epoch: every workgroup
// reads the epoch word at entry, the first GEMV workgroup writes epoch + 1 back when it is done -- by then every workgroup
// has read it (its own last wait depends on all of them) -- so a replayed hipGraph needs no memset node and a stale granule
// can only carry an older tag.  Every wait is bounded: on give-up an error word is set, results are garbage, nothing hangs.
//
// All NA + NG workgroups must be resident at once (<= one per CU: the host checks the CU count); the attention workgroups
// have the low block indices, are dispatched first and wait on nobody but the GEMV workgroups' first phase, which waits on
// nothing.
//
// Served: batch 1, bf16, int4 weights with one quantisation group per k-tile (g128), 16-bit cache, head size 128, <= 16
// query heads per KV group.  Everything else: dihip_decode_attn_block_supported() == 0 and the caller keeps the launch chain.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "gemv_stream_kernel.hpp"
#include "span_attn_ft_mfma.hpp"

namespace dihip {

bool gemv_block_plan(int wbits, int N, int K, int group_size, int nblocks, GemvArgs* g, int* max_units, size_t* lds_bytes);
void span_attn_block_plan(int batch, int n_heads, int n_groups, int max_seq_len, int* nsplits, int* nchunks, int* tps_static,
                          size_t* partial_bytes, int* waves);

constexpr int AB_THREADS = GEMV_THREADS;  // 8 waves
// The weight format of the launch's two GEMVs (both the same): what the stand-alone decode GEMV instantiates for it
//   WB = 4: int4, k-tiles of 128, ring of 8 chunks per wave;   WB = 8: int8, k-tiles of 64, ring of 16 chunks per wave
//   GPT   : a quantisation group per k-tile (int4 g128 -- the headline --, int8 g64): scale and zero applied per 1 KiB chunk
//   !GPT  : groups of several k-tiles (int8 g128, int4 g256 ...) or ONE group per column (per channel: InstantQuant int8, BASELINE
//           configs[1]; int4 per channel): the MFMA accumulators and the sum of x run through the group, one fma pair at its end
template <int WB>
struct AbFmt;
template <>
struct AbFmt<4> {
  static constexpr int KTILE = 128, KSTEPS = 4, RING = 8, EARLY_DEFAULT = 4;
};
template <>
struct AbFmt<8> {
  static constexpr int KTILE = 64, KSTEPS = 2, RING = 16, EARLY_DEFAULT = 8;
};
constexpr int AB_RING = AbFmt<4>::RING;   // 1 KiB weight chunks a wave holds per GEMV (int4): the whole share is resident
#ifndef DIHIP_AB_EARLY
#define DIHIP_AB_EARLY 4
#endif
constexpr int AB_EARLY = DIHIP_AB_EARLY;  // slots of the qkv share requested before the RMSNorm prologue (the rest at its barrier)


struct AttnBlockArgs {
  GemvArgs q;  // RMSNorm + qkv projection: x = f32 hidden row, gamma, eps, bias; output -> qkv_gran
  GemvArgs o;  // o projection: x <- out_gran; h_out = h_res + x . Wo
  AttnArgs a;
  unsigned* state;               // [0] epoch, [1] error, [2] record-buffer parity
  unsigned long long* qkv_gran;  // [(n + 2g) * H]
  unsigned long long* out_gran;  // [n * H / 2]
  size_t out_gran_bytes;
  unsigned* grp_flag;            // [g]
  unsigned* rec;                 // polled split records: two buffers of [n][nsplits][ATTN_PSTRIDE] words, zero before use
  unsigned rec_bytes;
  int NA, NG;                    // attention / GEMV workgroups
  unsigned spin_limit;
  unsigned fault;                // tests (DIHIP_ATTN_BLOCK_FAULT=1): GEMV workgroup 0 withholds its qkv rows -> the waits give up
  unsigned long long* trace;     // diagnostics (`make trace` build + dihip_debug_set_trace): [workgroup][32] wall-clock stamps, or null
};

// this wave's share of one GEMV (the bookkeeping of gemv_stream_body, M = 1)
struct AbShare {
  int wk, wn, k_lo, k_hi, nk, nu, total, u0, g_lo;
  const char* wtile;
  const char* stile;
  size_t wstep, sstep;
};

__device__ __forceinline__ AbShare ab_share(const GemvArgs& g, int bid, int NB, int wave) {
  AbShare s;
  const int lgWK = __builtin_ctz(g.WK), lgWN = __builtin_ctz(g.WN);
  s.wk = g.wmap ? wave >> lgWN : wave & (g.WK - 1);
  s.wn = g.wmap ? wave & (g.WN - 1) : wave >> lgWK;
  s.u0 = __builtin_amdgcn_readfirstlane(bid);
  s.nu = g.nu_q + (s.u0 < g.nu_r ? 1 : 0);
  const int gsz = g.ktpg < g.KT ? g.ktpg : 1;  // one group per k-tile (host contract): 1
  s.g_lo = __builtin_amdgcn_readfirstlane(g.kcut[s.wk]);
  s.k_lo = min(g.KT, s.g_lo * gsz);
  s.k_hi = min(g.KT, __builtin_amdgcn_readfirstlane(g.kcut[s.wk + 1]) * gsz);
  s.nk = s.k_hi - s.k_lo;
  const int nvw = s.wn < s.nu ? (s.nu - s.wn + g.WN - 1) >> lgWN : 0;
  s.total = __builtin_amdgcn_readfirstlane(nvw * s.nk);
  const int t0 = s.u0 + s.wn * NB, tstep = g.WN * NB;
  s.wtile = reinterpret_cast<const char*>(g.w0 + ((size_t)t0 * g.KT + s.k_lo) * 64);
  s.stile = reinterpret_cast<const char*>(g.sz0 + ((size_t)t0 * g.Gp + (g.ktpg < g.KT ? s.g_lo : 0)) * 16);  // (per-channel: the tile's one group)
  s.wstep = (size_t)tstep * g.KT * 1024;
  s.sstep = (size_t)tstep * g.Gp * 64;
  return s;
}

// the share is requested in the order the stand-alone kernel streams it, slots [J0, J1) per call (the cursor carries on).  Slots
// beyond the share re-read the head of the matrix: every asm load is unconditional, so no register of the ring is defined on one
// path only (a phi on an asm output may be resolved by a copy BEFORE the data has landed: tools/audit_asm_loads.py)
struct AbCursor {
  const char* wtile;
  const char* stile;
  const char* iwp;
  const char* isp;
  int ikt;
  int igl, gcount;  // !GPT: k-tiles left in the quantisation group / per group (per channel: never expires)
};
__device__ __forceinline__ AbCursor ab_cursor(const GemvArgs& g, const AbShare& s) {
  const int gcount = g.ktpg < g.KT ? g.ktpg : (1 << 30);
  return AbCursor{s.wtile, s.stile, s.wtile, s.stile, s.k_lo, gcount, gcount};
}
template <int J0, int J1, int RING, bool GPT>
__device__ __forceinline__ void ab_issue(const GemvArgs& g, const AbShare& s, AbCursor& c, u32x4_t (&wb)[RING], uint32_t (&sb)[RING],
                                         int lane) {
  const char* const dummy = reinterpret_cast<const char*>(g.w0);
  const uint32_t voff_w = (uint32_t)lane * 16u, voff_s = (uint32_t)(lane & 15) * 4u;
#pragma unroll
  for (int j = J0; j < J1; ++j) {
    const bool real = j < s.total;
    stream_load_b128(wb[j], uniform_ptr(real ? c.iwp : dummy), voff_w);
    stream_load_b32(sb[j], uniform_ptr(real ? c.isp : dummy), voff_s);
    c.iwp += 1024;
    if constexpr (GPT) {
      c.isp += 64;
    } else if (--c.igl == 0) {  // (every chunk of a group reads the group's one scale word; per channel: the tile's)
      c.igl = c.gcount;
      c.isp += 64;
    }
    if (++c.ikt == s.k_hi) {
      c.ikt = s.k_lo;
      c.igl = c.gcount;
      c.wtile += s.wstep;
      c.stile += s.sstep;
      c.iwp = c.wtile;
      c.isp = c.stile;
    }
  }
}

// per-k-tile sums of an 8-element vector of the staged row (the arithmetic and order of gemv_stream_body::stage_vector; KTILE 128 or 64)
template <int KTILE>
__device__ __forceinline__ void ab_stage_vector(uint16_t* xs, float* xsum_tab, int i, const u32x4_t& v, bool store) {
  if (store) *reinterpret_cast<u32x4_t*>(xs + (size_t)i * 8) = v;
  float e[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    e[2 * q] = bf16_bits_to_f32(v[q] & 0xFFFFu);
    e[2 * q + 1] = bf16_bits_to_f32(v[q] >> 16);
  }
  float sum = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
  sum += dpp_f32<0xB1>(sum);
  sum += dpp_f32<0x4E>(sum);
  sum += dpp_f32<0x141>(sum);
  if constexpr (KTILE >= 128) sum += dpp_f32<0x140>(sum);
  constexpr int VPT = KTILE / 8;  // vectors per k-tile
  if ((i & (VPT - 1)) == 0) xsum_tab[(i / VPT) * 16] = sum;
}

// the resident share against the staged row: DIHIP_GEMV_CONSUME of gemv_stream_body for bf16, M = 1 -- int4 with a group per k-tile (GPT:
// scale and zero per chunk) or int8 per channel (one scale / zero per column: the MFMA accumulators and the sum of x run through the
// wave's whole k-slice of a tile, one fma pair at its end)
template <int WB, bool GPT>
__device__ __forceinline__ void ab_consume(const GemvArgs& g, const AbShare& s, const u32x4_t (&wb)[AbFmt<WB>::RING],
                                           const uint32_t (&sb)[AbFmt<WB>::RING], unsigned char* smem, const float* xsum_tab, float* red, int lane) {
  using EX = ExpandV<WB, DIHIP_BF16>;
  constexpr int KTILE = AbFmt<WB>::KTILE, KSTEPS = AbFmt<WB>::KSTEPS, RING = AbFmt<WB>::RING;
  const int ni = lane & 15, kb = lane >> 4;
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  uint32_t ex_mask = 0x000F000Fu, ex_magic = 0x43004300u;
  asm volatile("" : "+v"(ex_mask), "+v"(ex_magic));
  if (s.nk == 0) {  // a k-slice without work: its partials must read as zero
    for (int v = s.wn; v < s.nu; v += g.WN) {
      float* dst = red + ((size_t)(v * g.WK + s.wk)) * 16;
      if (lane < 16) dst[lane] = 0.f;
    }
  }
  int cv = s.wn, ckt = s.k_lo;
  const bool arow_valid = ni < 1;
  const uint32_t xk_reset = arow_valid ? 256u + (uint32_t)(kb * 8 + s.k_lo * KTILE) * 2u : 0u;
  const uint32_t xk_step = arow_valid ? (uint32_t)KTILE * 2u : 0u;
  uint32_t xk = xk_reset;
  const float* xt = xsum_tab + s.k_lo * 16;
  float tot = 0.f;
  if constexpr (GPT) {
    auto chunk = [&](const u32x4_t& w, uint32_t sw) {
      f32x4_t g0 = zero4, g1 = zero4;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const u32x4_t af = *reinterpret_cast<const u32x4_t*>(smem + xk + ks * 64);
        const u32x4_t bf = EX::frag(w, ks, ex_mask, ex_magic);
        if (ks & 1) g1 = mfma16<DIHIP_BF16>(af, bf, ks == 1 ? zero4 : g1);
        else g0 = mfma16<DIHIP_BF16>(af, bf, ks == 0 ? zero4 : g0);
      }
      xk += xk_step;
      const bool tile_end = ++ckt == s.k_hi;
      const float s_ = bf16_bits_to_f32(sw & 0xFFFFu);
      const float nzp_ = -(bf16_bits_to_f32(sw >> 16) + EX::OFFSET);
      tot = fmaf(s_, fmaf(nzp_, xt[0], g0[0] + g1[0]), tot);
      xt += 16;
      if (tile_end) {
        float* dst = red + ((size_t)(cv * g.WK + s.wk)) * 16 + ni;
        if (kb == 0) dst[0] = tot;
        tot = 0.f;
        ckt = s.k_lo;
        cv += g.WN;
        xk = xk_reset;
        xt = xsum_tab + s.k_lo * 16;
      }
    };
#pragma unroll
    for (int j = 0; j < RING; ++j) {
      if constexpr (RING <= 8) {
        if (j >= s.total) break;
        chunk(wb[j], sb[j]);
      } else {  // (sixteen slots with a break do not unroll, and a ring indexed at run time lives in scratch)
        if (j < s.total) chunk(wb[j], sb[j]);
      }
    }
  } else {
    // groups of several k-tiles / per channel: the accumulators and the sum of x run through the group -- for a column's one group, through
    // the wave's k-slice of the tile (no early exit from the loop: sixteen slots with a break do not unroll, and a ring indexed at run
    // time lives in scratch)
    const int gcount = g.ktpg < g.KT ? g.ktpg : (1 << 30);
    int cgl = gcount;
    float xacc = 0.f;
    f32x4_t g0 = zero4, g1 = zero4;
#pragma unroll
    for (int j = 0; j < RING; ++j) {
      if (j < s.total) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const u32x4_t af = *reinterpret_cast<const u32x4_t*>(smem + xk + ks * 64);
          const u32x4_t bf = EX::frag(wb[j], ks, ex_mask, ex_magic);
          if (ks & 1) g1 = mfma16<DIHIP_BF16>(af, bf, g1);
          else g0 = mfma16<DIHIP_BF16>(af, bf, g0);
        }
        xk += xk_step;
        xacc += xt[0];
        xt += 16;
        const bool tile_end = ++ckt == s.k_hi;
        if (--cgl == 0 || tile_end) {  // group end (wave-uniform); a column's one group ends with the k-slice
          cgl = gcount;
          const float s_ = bf16_bits_to_f32(sb[j] & 0xFFFFu);
          const float nzp_ = -(bf16_bits_to_f32(sb[j] >> 16) + EX::OFFSET);
          tot = fmaf(s_, fmaf(nzp_, xacc, g0[0] + g1[0]), tot);
          xacc = 0.f;
          g0 = zero4;
          g1 = zero4;
        }
        if (tile_end) {
          float* dst = red + ((size_t)(cv * g.WK + s.wk)) * 16 + ni;
          if (kb == 0) dst[0] = tot;
          tot = 0.f;
          ckt = s.k_lo;
          cv += g.WN;
          xk = xk_reset;
          xt = xsum_tab + s.k_lo * 16;
        }
      }
    }
  }
}

// AW = live waves of an attention workgroup (the plan's: 4; 8 with DIHIP_ATTN_WIDE=1 -- the stand-alone kernel's forms, same records)
// WB = weight bits of the two GEMVs, GPT = a quantisation group per k-tile (AbFmt)
template <int AW, int WB = 4, bool GPT = true>
__global__ __launch_bounds__(AB_THREADS) void decode_attn_block_kernel(const AttnBlockArgs p) {
  constexpr int RING = AbFmt<WB>::RING, KTILE = AbFmt<WB>::KTILE;
  constexpr int EARLY = WB == 4 ? AB_EARLY : AbFmt<WB>::EARLY_DEFAULT;  // slots of the qkv share requested before the RMSNorm prologue
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bid = (int)blockIdx.x;
  unsigned tag = p.state[0] + 1u;
  if (tag == 0u) tag = 1u;

  // Workgroups are dispatched in the order of their ids, 256 of them over ~1.4 us: the GEMV workgroups take the FIRST ids -- the qkv rows
  // every attention workgroup waits for are published by the last of them -- the attention workgroups, whose K / V requests have ~3 us of
  // slack before q arrives, the last.  (-DDIHIP_AB_GEMV_FIRST=0: attention first, as rounds 5 / 6a had it.)
#ifndef DIHIP_AB_GEMV_FIRST
#define DIHIP_AB_GEMV_FIRST 1
#endif
  const int ab = DIHIP_AB_GEMV_FIRST ? bid - p.NG : bid;  // attention workgroup index, if >= 0 and < NA
  if (ab >= 0 && ab < p.NA) {
    // ------------------------------------------------------------------------------------------------ attention workgroup
    // (It takes no column tiles: its length / span-pointer / K / V loads are dependent round trips whose waits -- loads return
    // in order -- would also wait for a weight share requested before them, and a share requested after them would wait for
    // the 64 KB of K / V: either way the workgroup would publish its tiles late, and every workgroup waits for all tiles.)
    if (threadIdx.x >= AW * 64) return;  // an AW-wave body (barriers count live waves only)
    AttnHandoff ho;
    ho.qkv_gran = p.qkv_gran;
    ho.out_gran = p.out_gran;
    ho.out_gran_bytes = p.out_gran_bytes;
    ho.grp_flag = p.grp_flag;
    ho.err = p.state + 1;
    ho.tag = tag;
    ho.spin_limit = p.spin_limit;
    ho.rec = p.rec;
    ho.rec_bytes = p.rec_bytes;
    ho.parity = p.state[2] & 1u;
    const int ns = p.a.nsplits;
    span_attn_ft_mfma_body<DIHIP_BF16, DIHIP_KV_NONE, true, true, AW>(p.a, ab % ns, ab / ns, 0, ns, p.a.g, 1, smem, &ho);
    return;
  }

  // ---------------------------------------------------------------------------------------------------- GEMV workgroup
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const GemvArgs& q = p.q;
  const GemvArgs& o = p.o;
  const int NB = p.NG;
  const int lb = DIHIP_AB_GEMV_FIRST ? bid : bid - p.NA;
  const int ob = NB - 1 - lb;  // the o-projection's tiles are dealt in reverse block order: two qkv tiles -> one o tile
  // wave 0's wall-clock stamps (tools/attn_block_trace.py on the `make trace` build); absent from the product build
#if defined(DIHIP_GEMV_TRACE) && DIHIP_GEMV_TRACE
#define DIHIP_AB_STAMP(I)                                                              \
  do {                                                                                 \
    if (p.trace && wave == 0) p.trace[(size_t)(p.NA + lb) * 32 + (I)] = wall_clock64(); \
  } while (0)
#else
#define DIHIP_AB_STAMP(I) do { } while (0)
#endif
  DIHIP_AB_STAMP(0);

  // ---- loads that depend on nothing but the kernel arguments go out first: the epilogues' bias / residual elements (one per
  // thread: a dependent global load in an epilogue is ~1 us on the launch's critical path), then the early activation batch:
  // 8-element vectors i = j * THREADS + tid of the f32 hidden row + gamma (K <= 8192), then the qkv share ----
  const int nq_e = (q.nu_q + (lb < q.nu_r ? 1 : 0)) * 16, no_e = (o.nu_q + (ob < o.nu_r ? 1 : 0)) * 16;  // epilogue elements (<= 512)
  const int n_q = (lb + (tid >> 4) * NB) * 16 + (tid & 15), n_o = (ob + (tid >> 4) * NB) * 16 + (tid & 15);
  uint32_t bias_bits = 0u, hres_bits = 0u;
  {
    const bool bq = q.bias != nullptr && tid < nq_e && n_q < q.N, bo = o.h_res != nullptr && tid < no_e && n_o < o.N;
    // (unconditional asm loads, clamped: see ab_issue)
    asm volatile("global_load_ushort %0, %1, %2" : "=&v"(bias_bits) : "v"((uint32_t)(bq ? n_q : 0) * 2u), "s"(q.bias ? q.bias : q.gamma));
    asm volatile("global_load_dword %0, %1, %2" : "=&v"(hres_bits) : "v"((uint32_t)(bo ? n_o : 0) * 4u), "s"(o.h_res ? (const void*)o.h_res : q.x));
  }
  const int nvec_q = q.K >> 3;
  u32x4_t ev[4], eg[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    // (both batches unconditionally, indices clamped: a batch defined on one path only is a phi, and the compiler resolves a phi on
    // an asm load's registers with copies wherever it likes -- before the wait: tools/audit_asm_loads.py)
    const uint32_t i = (uint32_t)min(j * AB_THREADS + tid, nvec_q - 1);
    stream_load_plain_b128(ev[2 * j], q.x, i * 32u);
    stream_load_plain_b128(ev[2 * j + 1], q.x, i * 32u + 16u);
    stream_load_plain_b128(eg[j], q.gamma, i * 16u);
  }
  const AbShare sq = ab_share(q, lb, NB, wave);
  u32x4_t wq[RING], wo[RING];
  uint32_t sq_[RING], so_[RING];
  // the qkv share in two halves, AB_EARLY slots here and the rest at the RMSNorm's barrier: a CU holds ~32 KiB of outstanding misses,
  // and eight waves asking for 8 KiB each queue the second half of the workgroup behind the first (the stand-alone kernel's 4 + 4
  // ring fill, profiles/r03_gemv_wave_timeline.txt); -DDIHIP_AB_EARLY=8: everything at once (round 5)
  AbCursor cq = ab_cursor(q, sq);
  ab_issue<0, EARLY, RING, GPT>(q, sq, cq, wq, sq_, lane);

  uint16_t* xs = reinterpret_cast<uint16_t*>(smem + 256);
  float* xsum_q = reinterpret_cast<float*>(smem + 256 + gemv_xs_bytes(1, q.RS));
  float* red_q = xsum_q + (size_t)q.KT * 16;
  if (tid < 16) reinterpret_cast<u32x4_t*>(smem)[tid] = u32x4_t{0u, 0u, 0u, 0u};  // zero block (A rows >= M)

  // ---- RMSNorm prologue (gemv_stream_body PRO_RMSNORM, one row): rstd = 1/sqrt(mean(x^2)+eps); x_norm = FT((gamma*x)*rstd) ----
  stream_wait<2 * EARLY>();  // everything older than the ring's first part: the early batch
#pragma unroll
  for (int j = 0; j < 4; ++j) early_landed(ev[j]);
#pragma unroll
  for (int j = 0; j < 2; ++j) early_landed(eg[j]);
  asm volatile("" : "+v"(bias_bits), "+v"(hres_bits));  // (older than the early batch)
  {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j * AB_THREADS + tid < nvec_q) {
        const f32x4_t h0 = __builtin_bit_cast(f32x4_t, ev[2 * j]), h1 = __builtin_bit_cast(f32x4_t, ev[2 * j + 1]);
#pragma unroll
        for (int c = 0; c < 4; ++c) ss = fmaf(h0[c], h0[c], ss);
#pragma unroll
        for (int c = 0; c < 4; ++c) ss = fmaf(h1[c], h1[c], ss);
      }
    }
    ss = wave_sum(ss);
    if (lane == 0) red_q[wave] = ss;
    __syncthreads();
    ab_issue<EARLY, RING, RING, GPT>(q, sq, cq, wq, sq_, lane);  // (the rest of the qkv share flies while the row is normalised and staged)
    float tot_ss = 0.f;
#pragma unroll
    for (int w = 0; w < GEMV_WAVES; ++w) tot_ss += red_q[w];
    const float rstd = 1.f / sqrtf(tot_ss / (float)q.K + q.eps);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = j * AB_THREADS + tid;
      if (i < nvec_q) {
        const f32x4_t h0 = __builtin_bit_cast(f32x4_t, ev[2 * j]), h1 = __builtin_bit_cast(f32x4_t, ev[2 * j + 1]);
        u32x4_t v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float ga = bf16_bits_to_f32(eg[j][c] & 0xFFFFu), gb_ = bf16_bits_to_f32(eg[j][c] >> 16);
          const float xa = c < 2 ? h0[2 * c] : h1[2 * c - 4], xb = c < 2 ? h0[2 * c + 1] : h1[2 * c - 3];
          v[c] = pack_ft2<DIHIP_BF16>((ga * xa) * rstd, (gb_ * xb) * rstd);
        }
        ab_stage_vector<KTILE>(xs, xsum_q, i, v, true);
      }
    }
  }
  DIHIP_AB_STAMP(1);  // row normalised and staged
  stream_wait<0>();   // the qkv share has landed
#pragma unroll
  for (int j = 0; j < RING; ++j) stream_landed(wq[j], sq_[j]);
  __syncthreads();  // the row is staged; the RMS partials in red_q have been read
  DIHIP_AB_STAMP(2);  // qkv share landed

  // ---- qkv tiles of this workgroup ----
  ab_consume<WB, GPT>(q, sq, wq, sq_, smem, xsum_q, red_q, lane);
  __syncthreads();
  if (tid < nq_e && n_q < q.N && !(p.fault && lb == 0)) {  // element e = tid: column tile e / 16 of this workgroup, column e % 16
    float v = 0.f;
    const float* pr = red_q + ((size_t)(tid >> 4) * q.WK) * 16 + (tid & 15);
    for (int s = 0; s < q.WK; ++s) v += pr[(size_t)s * 16];
    v = __fmul_rn(q.alpha, v);
    if (q.bias) v = __fadd_rn(v, bf16_bits_to_f32(bias_bits & 0xFFFFu));
    const unsigned long long gran = ((unsigned long long)tag << 32) | (unsigned long long)f32_to_bf16_bits(v);
    __hip_atomic_store(p.qkv_gran + n_q, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  DIHIP_AB_STAMP(3);  // qkv tiles published
  // the o-projection's share: requested only now -- nothing on this workgroup's path needs it for several microseconds -- so
  // that it does not queue in front of the qkv shares every workgroup of the launch waits for
  const AbShare so = ab_share(o, ob, NB, wave);
  AbCursor co = ab_cursor(o, so);
  ab_issue<0, RING, RING, GPT>(o, so, co, wo, so_, lane);
  stream_wait<0>();
#pragma unroll
  for (int j = 0; j < RING; ++j) stream_landed(wo[j], so_[j]);

  float* xsum_o = reinterpret_cast<float*>(smem + 256 + gemv_xs_bytes(1, o.RS));
  float* red_o = xsum_o + (size_t)o.KT * 16;
  // ---- wait for the attention output, PER WAVE: a wave multiplies only its k-slice of the row (k-tiles [k_lo, k_hi) = vectors
  // [16 k_lo, 16 k_hi) of 8 elements), so it polls, sweeps and stages that slice itself and starts its tiles without a workgroup
  // barrier on either side (round 5: one polling wave, barrier, workgroup sweep, barrier: 2.5 us from the merged output to the end).
  // One lane per KV group polls the group's first output granule, then the slice's granules are swept until all carry this launch's
  // tag.  Waves sharing a k-slice (WN > 1) stage the same bytes twice. ----
  __syncthreads();  // every reader of red_q / xs of the first phase is done (far off the critical path: the output is microseconds away)
  for (unsigned spins = 0;; ++spins) {
    const unsigned f = lane < p.a.g ? (unsigned)(__hip_atomic_load(p.out_gran + (size_t)lane * p.a.hpg * 64, __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_AGENT) >> 32)
                                    : tag;
    if (__builtin_amdgcn_ballot_w64(f != tag) == 0ull) break;
    if (spins > p.spin_limit) {
      if (lane == 0) __hip_atomic_store(p.state + 1, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(4);
    if (wave & 1) __builtin_amdgcn_s_sleep(2);  // (the waves of a workgroup drift out of step)
  }
  DIHIP_AB_STAMP(4);  // every group's first output granule seen
  {
    const int v_lo = so.k_lo * (KTILE / 8), v_hi = so.k_hi * (KTILE / 8);  // whole k-tiles (16 or 8 lanes): the per-k-tile sums cross lanes
    const auto og_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out_gran, 0, (int)p.out_gran_bytes, 0x00020000);
    for (int i0 = v_lo; i0 < v_hi; i0 += 64) {
      const int i = i0 + lane, ic = min(i, v_hi - 1);
      // the vector's four granules = the two 16-byte stores {v01, tag, v23, tag} of the merging lanes: TWO 16-byte loads, not four of 8
      // (every load of a poll pass is on the launch's critical path: profiles/r06_attn_block_polls.txt)
      u32x4_t g0, g1;
      for (unsigned spins = 0;; ++spins) {
        g0 = __builtin_amdgcn_raw_buffer_load_b128(og_rsrc, (uint32_t)ic * 32u, 0, 16 /* sc1 */);
        g1 = __builtin_amdgcn_raw_buffer_load_b128(og_rsrc, (uint32_t)ic * 32u + 16u, 0, 16);
        const bool ok = g0[1] == tag && g0[3] == tag && g1[1] == tag && g1[3] == tag;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        if (spins > p.spin_limit) {
          if (lane == 0) __hip_atomic_store(p.state + 1, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      const u32x4_t v = {g0[0], g0[2], g1[0], g1[2]};
      ab_stage_vector<KTILE>(xs, xsum_o, ic, v, i < v_hi);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave reads back what it staged itself: LDS operations of a wave execute in order
  __builtin_amdgcn_wave_barrier();
  DIHIP_AB_STAMP(5);  // this wave's slice of the attention output swept into LDS

  // ---- o-projection tiles: h_out = h_res + attn . Wo ----
  ab_consume<WB, GPT>(o, so, wo, so_, smem, xsum_o, red_o, lane);
  __syncthreads();
  DIHIP_AB_STAMP(6);  // o tiles multiplied
  if (tid < no_e && n_o < o.N) {
    float v = 0.f;
    const float* pr = red_o + ((size_t)(tid >> 4) * o.WK) * 16 + (tid & 15);
    for (int s = 0; s < o.WK; ++s) v += pr[(size_t)s * 16];
    const float base = o.h_res ? __uint_as_float(hres_bits) : 0.f;
    o.h_out[n_o] = __fadd_rn(base, __fmul_rn(o.alpha, v));
  }
  DIHIP_AB_STAMP(7);
#undef DIHIP_AB_STAMP
  // the next launch's epoch (see the header: every workgroup has read the old one by now)
  if (lb == 0 && tid == 0) {
    p.state[0] = tag;
    p.state[2] = (p.state[2] & 1u) ^ 1u;  // record-buffer parity of the next launch (its own word: the tag skips 0 when it wraps)
  }
}

static bool attn_block_enabled() {
  static const bool on = !env_off("DIHIP_ATTN_BLOCK");  // =0: "not supported" (callers keep the launch chain; A/B)
  return on;
}

struct AbLayout {
  size_t flags, tickets, qkv_gran, out_gran, rec, rec_bytes, total;
};
constexpr int AB_POLLED_MAX_SPLITS = 32;  // polled split records: one load batch of the merger covers every split
// attention / GEMV workgroup counts: every workgroup resident (<= one per CU), every GEMV workgroup owns >= 1 tile of both matrices
static bool ab_grid(int n_heads, int n_groups, int head_size, int hidden, int nsplits, int* NA, int* NG) {
  const int ncu = cached_num_cus();
  if (ncu <= 0) return false;
  *NA = nsplits * n_groups;
  *NG = std::min(ncu - *NA, std::min((n_heads + 2 * n_groups) * head_size / 16, hidden / 16));
  return *NG >= 32;
}
using AbKernel = void (*)(const AttnBlockArgs);
static AbKernel ab_kernel(int aw, int wbits, bool gpt) {
  if (wbits == 8) {
    if (gpt) return aw == 8 ? decode_attn_block_kernel<8, 8, true> : decode_attn_block_kernel<4, 8, true>;
    return aw == 8 ? decode_attn_block_kernel<8, 8, false> : decode_attn_block_kernel<4, 8, false>;
  }
  if (gpt) return aw == 8 ? decode_attn_block_kernel<8, 4, true> : decode_attn_block_kernel<4, 4, true>;
  return aw == 8 ? decode_attn_block_kernel<8, 4, false> : decode_attn_block_kernel<4, 4, false>;
}
static size_t ab_attn_lds(int aw) { return (size_t)((ft_mfma_smem_bytes(aw) + 15) & ~15) + FT_MFMA_GATHER_IMG_BYTES; }
static AbLayout ab_layout(int n_heads, int n_groups, int head_size) {
  AbLayout l;
  l.flags = 64;
  l.tickets = 128;
  l.qkv_gran = (l.tickets + (size_t)n_groups * 128 + 255) & ~(size_t)255;
  l.out_gran = l.qkv_gran + (size_t)(n_heads + 2 * n_groups) * head_size * 8;
  l.rec = (l.out_gran + (size_t)n_heads * head_size / 2 * 8 + 255) & ~(size_t)255;
  l.rec_bytes = (size_t)n_heads * AB_POLLED_MAX_SPLITS * ATTN_PSTRIDE * sizeof(float);  // one buffer; two alternate by launch parity
  l.total = l.rec + 2 * l.rec_bytes;
  return l;
}

// The record layout (heads x splits) the polled records of `sync` were last used with.  A change clears the record region on `stream`
// -- eagerly: a memset captured into a hipGraph would run at every replay, and was observed to corrupt replayed steps (round 6: the
// C++ runner's first captured step after a layout change, tests/test_gpu_attn_block.py::test_launch_plans_...); under capture a change
// is an error that names the call to make first.
static int ab_sync_layout(hipStream_t stream, const void* sync, const AbLayout& lay, int n_heads, int ns) {
  static std::mutex mu;
  static std::unordered_map<const void*, unsigned> last_layout;
  const unsigned key = ((unsigned)n_heads << 16) | (unsigned)ns;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = last_layout.find(sync);
    if (it == last_layout.end()) {  // first use: the caller zeroed the buffer (contract)
      last_layout[sync] = key;
      return DIHIP_SUCCESS;
    }
    if (it->second == key) return DIHIP_SUCCESS;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
  (void)hipGetLastError();
  DIHIP_REQUIRE(cs == hipStreamCaptureStatusNone, DIHIP_RUNTIME_ERROR,
                "decode_attn_block: the split plan of this sync buffer changed inside a stream capture -- call dihip_decode_attn_block_prepare() "
                "(or run one launch) outside the capture first");
  char* sb = const_cast<char*>(reinterpret_cast<const char*>(sync));
  DIHIP_CHECK_HIP(hipMemsetAsync(sb + lay.rec, 0, 2 * lay.rec_bytes, stream), DIHIP_RUNTIME_ERROR);
  std::lock_guard<std::mutex> lk(mu);
  last_layout[sync] = key;
  return DIHIP_SUCCESS;
}

}  // namespace dihip

using namespace dihip;

extern "C" {

int dihip_decode_attn_block_prepare(void* stream, void* sync, size_t sync_bytes, int n_heads, int n_groups, int head_size, int max_seq_len) {
  DIHIP_REQUIRE(sync && n_heads > 0 && n_groups > 0 && n_heads % n_groups == 0 && max_seq_len > 0, DIHIP_PARAM_ERROR, "decode_attn_block_prepare: bad argument");
  const AbLayout lay = ab_layout(n_heads, n_groups, head_size);
  DIHIP_REQUIRE(sync_bytes >= lay.total, DIHIP_MEMORY_ERROR, "decode_attn_block_prepare: sync buffer too small (%zu < %zu)", sync_bytes, lay.total);
  int ns, nc, tps, aw;
  size_t pb;
  span_attn_block_plan(1, n_heads, n_groups, max_seq_len, &ns, &nc, &tps, &pb, &aw);
  return ab_sync_layout(reinterpret_cast<hipStream_t>(stream), sync, lay, n_heads, ns);
}

int dihip_decode_attn_block_supported(int wbits, int group_size, int hidden, int n_heads, int n_groups, int head_size, int max_seq_len,
                                      int kv_mode, int dtype, int batch) {
  if (!attn_block_enabled() || batch != 1 || (wbits != 4 && wbits != 8) || dtype != DIHIP_BF16 || kv_mode != DIHIP_KV_NONE || head_size != 128) return 0;
  static const bool w8_on = !env_off("DIHIP_ATTN_BLOCK_W8");  // =0: int8 weights keep the three launches (A/B)
  if (wbits == 8 && !w8_on) return 0;
  if (n_heads <= 0 || n_groups <= 0 || n_groups > 16 || n_heads % n_groups || n_heads / n_groups > MF_HC || hidden <= 0 || hidden > 8192 ||
      hidden % 128 || max_seq_len <= 0)
    return 0;
  int ns, nc, tps, aw;
  size_t pb;
  span_attn_block_plan(1, n_heads, n_groups, max_seq_len, &ns, &nc, &tps, &pb, &aw);
  if (nc != 1 || ns < 2 || ns > AB_POLLED_MAX_SPLITS || pb >= (1ull << 31)) return 0;  // (one split: no merge -- the chain's single launch is as good; > 32: the record buffers' layout)
  int NA, NG;
  if (!ab_grid(n_heads, n_groups, head_size, hidden, ns, &NA, &NG)) return 0;
  GemvArgs gq{}, go{};
  int mu;
  size_t lds;
  const int Nq = (n_heads + 2 * n_groups) * head_size, Ko = n_heads * head_size;
  // a group per k-tile (GPT form), groups of whole k-tiles, or one group per column -- both matrices the same form
  auto fmt_ok = [&](const GemvArgs& g) { return g.ktpg >= 1; };
  size_t lds_o = 0;
  if (!gemv_block_plan(wbits, Nq, hidden, group_size, NG, &gq, &mu, &lds) || !fmt_ok(gq)) return 0;
  if (!gemv_block_plan(wbits, hidden, Ko, group_size, NG, &go, &mu, &lds_o) || !fmt_ok(go) || Ko > 8192 || (gq.ktpg == 1) != (go.ktpg == 1)) return 0;
  lds = std::max(lds, lds_o);
  // every wave's share must fit its ring: ceil(units / WN) * longest k-slice
  const int ring = wbits == 4 ? AbFmt<4>::RING : AbFmt<8>::RING;
  auto fits = [ring](const GemvArgs& g) {
    int longest = 0;
    const int gsz = g.ktpg < g.KT ? g.ktpg : 1;
    for (int i = 0; i < g.WK; ++i) longest = std::max(longest, std::min(g.KT, g.kcut[i + 1] * gsz) - std::min(g.KT, g.kcut[i] * gsz));
    return ((g.upb + g.WN - 1) / g.WN) * longest <= ring;
  };
  if (!fits(gq) || !fits(go)) return 0;
  // every workgroup of the launch must be RESIDENT at once (they wait for one another): the grid is <= one workgroup per CU by
  // construction; the kernel itself must fit a CU with its LDS at this block size (ask the occupancy calculator, once)
  const size_t lds_need = std::max<size_t>(lds, ab_attn_lds(aw));
  static std::atomic<int> occ[4] = {{-1}, {-1}, {-1}, {-1}};
  std::atomic<int>& oc = occ[(aw == 8) + 2 * (wbits == 8)];  // (the GPT and the group forms of a width differ in a few VALU instructions: one answer)
  int o = oc.load(std::memory_order_relaxed);
  if (o < 0) {
    int nb = 0;
    const size_t lds_q = std::max<size_t>(lds_need, 96 * 1024);  // (asked with a generous LDS figure: the answer is cached for all shapes)
    const auto kern = ab_kernel(aw, wbits, gq.ktpg == 1);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q) != hipSuccess)
      nb = 0;
    else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, AB_THREADS, lds_q) != hipSuccess)
      nb = 0;
    (void)hipGetLastError();
    o = nb;
    oc.store(o, std::memory_order_relaxed);
  }
  return o >= 1 && lds_need <= 150 * 1024 ? 1 : 0;
}

size_t dihip_decode_attn_block_sync_bytes(int n_heads, int n_groups, int head_size) {
  if (n_heads <= 0 || n_groups <= 0 || head_size <= 0) return 0;
  return ab_layout(n_heads, n_groups, head_size).total;
}

size_t dihip_decode_attn_block_workspace_bytes(int n_heads, int n_groups, int head_size, int max_seq_len) {
  if (n_heads <= 0 || n_groups <= 0 || n_heads % n_groups || max_seq_len <= 0) return 0;
  (void)head_size;
  int ns, nc, tps, aw;
  size_t pb;
  span_attn_block_plan(1, n_heads, n_groups, max_seq_len, &ns, &nc, &tps, &pb, &aw);
  return pb + 256;
}

int dihip_decode_attn_block_status_async(void* stream, const void* sync, unsigned* host_word) {
  DIHIP_REQUIRE(sync && host_word, DIHIP_PARAM_ERROR, "decode_attn_block_status: null pointer");
  DIHIP_CHECK_HIP(hipMemcpyAsync(host_word, reinterpret_cast<const unsigned*>(sync) + 1, sizeof(unsigned), hipMemcpyDeviceToHost,
                                 reinterpret_cast<hipStream_t>(stream)),
                  DIHIP_RUNTIME_ERROR);
  return DIHIP_SUCCESS;
}

int dihip_decode_attn_block_reset(void* stream, void* sync, size_t sync_bytes) {
  DIHIP_REQUIRE(sync && sync_bytes >= 64, DIHIP_PARAM_ERROR, "decode_attn_block_reset: bad argument");
  DIHIP_CHECK_HIP(hipMemsetAsync(sync, 0, sync_bytes, reinterpret_cast<hipStream_t>(stream)), DIHIP_RUNTIME_ERROR);
  return DIHIP_SUCCESS;
}

int dihip_decode_attn_block(void* stream, int wbits, const float* h_in, const float* h_res, float* h_out, const void* gamma, float eps,
                            const void* qkv_w, const void* qkv_sz, const void* qkv_bias, const void* o_w, const void* o_sz,
                            void* const* k_span_array, void* const* v_span_array, const uint32_t* old_seq_lens_dev,
                            const float* rope_table, int hidden, int n_heads, int n_groups, int head_size, int group_size, int span_len,
                            int n_spans_per_request, int max_seq_len, int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes,
                            void* sync, size_t sync_bytes) {
  DIHIP_REQUIRE(h_in && h_out && gamma && qkv_w && qkv_sz && o_w && o_sz && k_span_array && v_span_array && old_seq_lens_dev && rope_table &&
                    ws && sync,
                DIHIP_PARAM_ERROR, "decode_attn_block: null pointer");
  DIHIP_REQUIRE(span_len == 16 || span_len == 32 || span_len == 64 || span_len == 128, DIHIP_PARAM_ERROR,
                "span_attn: span length %d not in {16,32,64,128}", span_len);
  DIHIP_REQUIRE(n_spans_per_request > 0, DIHIP_PARAM_ERROR, "decode_attn_block: invalid parameter");
  DIHIP_REQUIRE(dihip_decode_attn_block_supported(wbits, group_size, hidden, n_heads, n_groups, head_size, max_seq_len, kv_mode, dtype, 1),
                DIHIP_PARAM_ERROR,
                "decode_attn_block: configuration not covered (batch 1, bf16, int4 / int8 weights, 16-bit cache, head size 128); see _supported");
  DIHIP_REQUIRE(reinterpret_cast<uintptr_t>(h_in) % 16 == 0 && reinterpret_cast<uintptr_t>(gamma) % 16 == 0, DIHIP_PARAM_ERROR,
                "decode_attn_block: the hidden row and gamma must be 16-byte aligned");
  const AbLayout lay = ab_layout(n_heads, n_groups, head_size);
  DIHIP_REQUIRE(sync_bytes >= lay.total && reinterpret_cast<uintptr_t>(sync) % 16 == 0, DIHIP_MEMORY_ERROR,
                "decode_attn_block: sync buffer too small (%zu < %zu)", sync_bytes, lay.total);
  int ns, nc, tps, aw;
  size_t pb;
  span_attn_block_plan(1, n_heads, n_groups, max_seq_len, &ns, &nc, &tps, &pb, &aw);
  DIHIP_REQUIRE(ws_bytes >= pb, DIHIP_MEMORY_ERROR, "decode_attn_block: workspace too small (%zu < %zu)", ws_bytes, pb);
  AttnBlockArgs p{};
  ab_grid(n_heads, n_groups, head_size, hidden, ns, &p.NA, &p.NG);
  int mu;
  size_t lds_q, lds_o;
  const int Nq = (n_heads + 2 * n_groups) * head_size, Ko = n_heads * head_size;
  gemv_block_plan(wbits, Nq, hidden, group_size, p.NG, &p.q, &mu, &lds_q);
  gemv_block_plan(wbits, hidden, Ko, group_size, p.NG, &p.o, &mu, &lds_o);
  p.q.w0 = reinterpret_cast<const u32x4_t*>(qkv_w);
  p.q.sz0 = reinterpret_cast<const uint32_t*>(qkv_sz);
  p.q.x = h_in;
  p.q.gamma = gamma;
  p.q.eps = eps;
  p.q.bias = qkv_bias;
  p.o.w0 = reinterpret_cast<const u32x4_t*>(o_w);
  p.o.sz0 = reinterpret_cast<const uint32_t*>(o_sz);
  p.o.h_res = h_res;
  p.o.h_out = h_out;
  char* sb = reinterpret_cast<char*>(sync);
  p.state = reinterpret_cast<unsigned*>(sb);
  p.grp_flag = reinterpret_cast<unsigned*>(sb + lay.flags);
  p.qkv_gran = reinterpret_cast<unsigned long long*>(sb + lay.qkv_gran);
  p.out_gran = reinterpret_cast<unsigned long long*>(sb + lay.out_gran);
  p.out_gran_bytes = (size_t)n_heads * head_size / 2 * 8;
  // split records polled by their items' owners (the only protocol of the block since round 6; _supported() admits ns <= 32)
  {
    p.rec = reinterpret_cast<unsigned*>(sb + lay.rec);
    p.rec_bytes = (unsigned)((size_t)n_heads * ns * ATTN_PSTRIDE * sizeof(float));
    // the record buffers are zero between launches only where the LAST launch's owners zeroed them: a launch with another record
    // layout on the same sync buffer (another split count) starts from a cleared region -- ab_sync_layout, NEVER inside a stream
    // capture (see dihip_decode_attn_block_prepare)
    const int st = ab_sync_layout(reinterpret_cast<hipStream_t>(stream), sync, lay, n_heads, ns);
    if (st != DIHIP_SUCCESS) return st;
  }
  static const unsigned spin_limit = (unsigned)std::max(1024, env_int("DIHIP_ATTN_BLOCK_SPINS", 1 << 18));
  p.spin_limit = spin_limit;
  {
    const char* f = getenv("DIHIP_ATTN_BLOCK_FAULT");  // (read per call: a test switches it inside one process)
    p.fault = f && f[0] == '1';
    if (p.fault) p.spin_limit = 1024;
  }
  AttnArgs& a = p.a;
  a.kspans = k_span_array;
  a.vspans = v_span_array;
  a.seq_lens = old_seq_lens_dev;
  a.partials = reinterpret_cast<float*>(ws);
  a.counters = reinterpret_cast<unsigned*>(sb + lay.tickets);
  a.B = 1;
  a.n = n_heads;
  a.g = n_groups;
  a.hpg = n_heads / n_groups;
  a.S = span_len;
  a.span_stride = n_spans_per_request;
  a.nsplits = ns;
  a.nchunks = 1;
  a.scale = qk_scale;
  a.rope_tab = rope_table;
  a.tps_static = tps;
  a.merge_wt = 1;
  a.partial_bytes = pb;
  p.trace = debug_trace_buffer((size_t)(p.NA + p.NG) * 32 * sizeof(unsigned long long));
  a.trace = p.trace;  // (the attention body's own stamps: [workgroup][wave][8])
  const size_t lds = std::max<size_t>(std::max(lds_q, lds_o), ab_attn_lds(aw));
  const bool gpt = p.q.ktpg == 1;
  const auto kern = ab_kernel(aw, wbits, gpt);
  if (lds > 64 * 1024) {
    static std::atomic<size_t> granted[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
    std::atomic<size_t>& gr = granted[(aw == 8) + 2 * (wbits == 8) + 4 * gpt];
    if (lds > gr.load(std::memory_order_relaxed)) {
      DIHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                      DIHIP_RUNTIME_ERROR);
      gr.store(lds, std::memory_order_relaxed);
    }
  }
  hipLaunchKernelGGL(kern, dim3(p.NA + p.NG), dim3(AB_THREADS), lds, reinterpret_cast<hipStream_t>(stream), p);
  return launch_status();
}

}  // extern "C"
