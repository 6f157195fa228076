// span_attn.hip -- paged-KV decode attention for gfx950 (include/dashinfer_hip.h section 3).
//
// Replaces the span-attention library's 3-5 kernel pipeline (QKGemv -> softmax -> QKVGemv ->
// reduce, span-attention/src/attn/span_attention.hpp:145-211, with FT scores between kernels and
// per-step host-built tile maps) by one single-pass kernel + a small split-merge launch.  Common to all
// kernels here: grid = (kv-splits, kv-groups x head-chunks, requests); every workgroup streams a contiguous
// token range of one request's K and V spans exactly once, every KV row is reused by all query heads of its
// GQA group, scores / online softmax / P.V accumulation stay in f32, split partials (m, l, o) are merged by
// span_attn_split_merge_kernel (cheaper than an in-kernel last-arriver hand-off, whose agent-scope fences
// cost ~20 us per layer at batch 32), and sequence lengths are read on the device: no host-side handle /
// tile-map rebuild per step.
//   span_attn_decode_kernel        VALU, every dtype x cache mode (16 lanes x 16 B cover a token-head row; QK
//                                  rows reduced over the 16 lanes with DPP rotations; 8 heads per workgroup).
//                                  Still used for f32 activations and f16 + uint4.
//   span_attn_u4_mfma_kernel       uint4 cache, bf16 activations: S^T = K.Q^T and O^T = V^T.P^T on MFMA with an
//                                  in-register nibble transpose for V^T.
//   span_attn_ft_mfma_kernel       16-bit and int8 caches on MFMA (V^T through an LDS tile + transpose reads);
//                                  FUSED = the decode-step form with Rotary and the cache append folded in.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <new>
#include <type_traits>
#include <vector>

#include "span_attn_common.hpp"
#include "span_attn_ft_mfma.hpp"
#include "span_codec.hpp"

namespace dihip {

// One workgroup (128 threads = head dims) per (request, head): combines the nsplits partial records.  Splits
// beyond the sequence hold neutral records (m = -inf, l = 0), so no length is needed.
template <int FT>
__global__ __launch_bounds__(128) void span_attn_split_merge_kernel(void* out, const float* partials, int n, int nsplits,
                                                                    int out_frag_mt) {
  constexpr int H = 128;
  const int bh = blockIdx.x, d = threadIdx.x;
  const float* base = partials + (size_t)bh * nsplits * ATTN_PSTRIDE;
  constexpr int MB = 32;  // splits per batch: all loads of a batch are in flight together (one round trip up to 32 splits:
                          // the 17-split plan of batch 1 at 2048 tokens took two with MB = 16, ~1.3 us of a 4.5 us launch)
  // merge_order4 (span_attn_common.hpp): the maximum of all splits first (beyond one batch: a pass over the m words) ...
  float M = -INFINITY;
  if (nsplits > MB) {
    for (int sb = 0; sb < nsplits; sb += MB) {
      float mv[MB];
#pragma unroll
      for (int j = 0; j < MB; ++j) mv[j] = base[(size_t)min(sb + j, nsplits - 1) * ATTN_PSTRIDE + H];
#pragma unroll
      for (int j = 0; j < MB; ++j) M = fmaxf(M, mv[j]);
    }
  }
  // ... then four fma chains over j = r (mod 4), combined as (s0 + s1) + (s2 + s3)
  float l4[4] = {0.f, 0.f, 0.f, 0.f}, o4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int sb = 0; sb < nsplits; sb += MB) {
    float mv[MB], lv[MB], ov[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const float* rec = base + (size_t)min(sb + j, nsplits - 1) * ATTN_PSTRIDE;
      mv[j] = rec[H];
      lv[j] = rec[H + 1];
      ov[j] = rec[d];
    }
    if (nsplits <= MB) {
#pragma unroll
      for (int j = 0; j < MB; ++j) M = fmaxf(M, mv[j]);
    }
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      if (sb + j < nsplits) {
        const float c = safe_exp_diff(mv[j], M);
        l4[j & 3] = fmaf(lv[j], c, l4[j & 3]);
        o4[j & 3] = fmaf(ov[j], c, o4[j & 3]);
      }
    }
  }
  const float ll = (l4[0] + l4[1]) + (l4[2] + l4[3]), oo = (o4[0] + o4[1]) + (o4[2] + o4[3]);
  const int b = bh / n, h = bh - b * n;
  const size_t idx = out_frag_mt ? act_frag_index(b, h * H + d, out_frag_mt) : (size_t)bh * H + d;
  store_ft<FT>(out, idx, ll > 0.f ? oo / ll : 0.f);
}

template <int FT, int MODE, int HC>
__global__ __launch_bounds__(ATTN_THREADS) void span_attn_decode_kernel(const AttnArgs a) {
  constexpr int H = 128;
  constexpr int TB = ATTN_TB;  // tokens per lane-slot per iteration
  __shared__ __attribute__((aligned(16))) float lds[4 * HC * ATTN_PSTRIDE + 4];
  unsigned* flag_lds = reinterpret_cast<unsigned*>(lds + 4 * HC * ATTN_PSTRIDE);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tl = lane >> 4, dc = lane & 15;
  const int split = blockIdx.x;
  const int grp = blockIdx.y / a.nchunks, hc = blockIdx.y % a.nchunks;
  const int b = blockIdx.z;
  const int h0 = grp * a.hpg + hc * HC;
  const int nh = min(HC, a.hpg - hc * HC);

  const int len = (int)a.seq_lens[b] + a.len_bias;
  const int tps = ((len + a.nsplits - 1) / a.nsplits + 15) & ~15;
  const int t0 = split * tps;
  const int t1 = min(len, t0 + tps);

  const void* const* ksp = a.kspans + (size_t)b * a.span_stride;
  const void* const* vsp = a.vspans + (size_t)b * a.span_stride;

  // token of (iteration base tb, slot i) for this lane: 16 consecutive tokens per wave-load group
  auto issue = [&](KvChunk<FT, MODE> (&kc)[TB], KvChunk<FT, MODE> (&vc)[TB], int tb) {
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      // a wave-load row = 4 consecutive tokens starting at a multiple of 4: one span (span lengths are
      // multiples of 16), so the span pointers are wave-uniform scalar loads -- a per-lane vector load of the
      // pointer would queue behind the prefetched rows (in-order vmcnt) and drain them
      int rb = tb + i * 16 + wave * 4;
      rb = rb < t1 ? rb : t0;  // a row past the range re-reads the first row (legal address, result discarded)
      const int sp = __builtin_amdgcn_readfirstlane(rb / a.S);
      const int tt = min(rb + tl, t1 - 1);  // stays inside the row: same span
      const int pos = tt - sp * a.S;
      kv_issue<FT, MODE>(kc[i], ksp[sp], grp, pos, a.g, a.S, dc);
      kv_issue<FT, MODE>(vc[i], vsp[sp], grp, pos, a.g, a.S, dc);
    }
  };
  KvChunk<FT, MODE> k0[TB], v0[TB], k1[TB], v1[TB];
  if (t0 < t1) issue(k0, v0, t0);  // first tokens in flight before q is touched

  // q (pre-scaled) for this lane's 8 dims of every head of the chunk
  float qr[HC][8];
#pragma unroll
  for (int h = 0; h < HC; ++h) {
#pragma unroll
    for (int j = 0; j < 8; ++j) qr[h][j] = 0.f;
    if (h < nh) {
      const size_t off = ((size_t)b * a.n + h0 + h) * H + dc * 8;
      if constexpr (FT == DIHIP_F32) {
        const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(a.q) + off);
        const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(a.q) + off + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          qr[h][j] = a.scale * q0[j];
          qr[h][4 + j] = a.scale * q1[j];
        }
      } else {
        const u32x4_t qv = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(a.q) + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          qr[h][2 * j] = a.scale * ft_bits_to_f32<FT == DIHIP_F32 ? DIHIP_BF16 : FT>(qv[j] & 0xFFFFu);
          qr[h][2 * j + 1] = a.scale * ft_bits_to_f32<FT == DIHIP_F32 ? DIHIP_BF16 : FT>(qv[j] >> 16);
        }
      }
    }
  }
  float m[HC], l[HC], o[HC][8];
#pragma unroll
  for (int h = 0; h < HC; ++h) {
    m[h] = -INFINITY;
    l[h] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[h][j] = 0.f;
  }

  auto process = [&](const KvChunk<FT, MODE> (&kc)[TB], const KvChunk<FT, MODE> (&vc)[TB], int tb) {
    float s[TB][HC];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const bool valid = tb + i * 16 + wave * 4 + tl < t1;
      float kx[8];
      kv_decode<FT, MODE>(kc[i], kx);
#pragma unroll
      for (int h = 0; h < HC; ++h) {
        float p = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) p = fmaf(qr[h][j], kx[j], p);
        p = row16_sum(p);
        s[i][h] = valid ? p : -INFINITY;
      }
    }
#pragma unroll
    for (int h = 0; h < HC; ++h) {
      float mn = m[h];
#pragma unroll
      for (int i = 0; i < TB; ++i) mn = fmaxf(mn, s[i][h]);
      const float corr = safe_exp_diff(m[h], mn);
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        s[i][h] = safe_exp_diff(s[i][h], mn);  // now the softmax weight
        ps += s[i][h];
      }
      l[h] = l[h] * corr + ps;
      m[h] = mn;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[h][j] *= corr;
    }
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      float vx[8];
      kv_decode<FT, MODE>(vc[i], vx);
#pragma unroll
      for (int h = 0; h < HC; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[h][j] = fmaf(s[i][h], vx[j], o[h][j]);
    }
  };

  // double-buffered token stream: the next iteration's K/V rows are in flight while this one
  // is reduced
  constexpr int STEP = ATTN_TOK_PER_ITER;
  for (int tb = t0; tb < t1; tb += 2 * STEP) {
    // unconditional (rows past the range are clamped inside issue): a load behind a branch makes hipcc
    // drain the whole queue at the join (s_waitcnt vmcnt(0)), which would serialise loads and arithmetic
    issue(k1, v1, tb + STEP);
    process(k0, v0, tb);
    issue(k0, v0, tb + 2 * STEP);
    if (tb + STEP < t1) process(k1, v1, tb + STEP);
  }

  // ---- merge the 4 token slots of the wave (lanes with equal dc) ---------------------------
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
    for (int h = 0; h < HC; ++h) {
      const float mo = __shfl_xor(m[h], off, 64), lo = __shfl_xor(l[h], off, 64);
      const float mn = fmaxf(m[h], mo);
      const float ca = safe_exp_diff(m[h], mn), cb = safe_exp_diff(mo, mn);
      l[h] = l[h] * ca + lo * cb;
      m[h] = mn;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[h][j] = o[h][j] * ca + __shfl_xor(o[h][j], off, 64) * cb;
    }
  }
  // ---- merge the 4 waves through LDS ---------------------------------------------------------
  if (tl == 0) {
#pragma unroll
    for (int h = 0; h < HC; ++h) {
      float* rec = lds + (wave * HC + h) * ATTN_PSTRIDE;
      *reinterpret_cast<f32x4_t*>(rec + dc * 8) = f32x4_t{o[h][0], o[h][1], o[h][2], o[h][3]};
      *reinterpret_cast<f32x4_t*>(rec + dc * 8 + 4) = f32x4_t{o[h][4], o[h][5], o[h][6], o[h][7]};
      if (dc == 0) {
        rec[H] = m[h];
        rec[H + 1] = l[h];
      }
    }
  }
  attn_block_epilogue<FT, HC>(a, lds, flag_lds, b, h0, nh, split);
}

// ---- uint4 KV cache, 16-bit bf16 activations: both contractions on the matrix cores ---------------------
// The VALU kernel above spends ~45 instructions per (token, head group) on a u4 cache (4-byte loads, nibble
// decode, 16-lane reductions) and keeps only a few KB per CU in flight.  Here a wave takes 32 tokens per
// iteration:
//   S^T[token, head] = K[token, :] . Q[head, :]   16x16x32 MFMA, A = K rows exactly as the span stores them
//       (lane (kb, token) <- the 16 bytes = 32 dims kb*32.. of the token row: one 1 KiB wave-load per 16
//       tokens; a dword expands to 8 exact bf16 integers 128+q with 7 VALU ops), B = Q (bf16, unscaled);
//       zero-point and scales are applied to the f32 scores: s*alpha*(acc - (128+z)*sum_d q);
//   transposed scores put a head in a lane (ni) and 4 tokens per 16-lane row (kb): the online softmax needs
//       two cross-row shuffles per iteration and P is already in the B-operand layout of
//   O^T[dim, head] += V^T[dim, token] . P'[token, head]   (P' = P * v_scale, rounded to bf16).  A = V^T comes
//       from 8 dword loads per lane (dword `ni` of 8 token rows) and an in-register 8x8 nibble transpose
//       (v_perm_b32); the k-slot <-> token and row <-> dim maps are free, so no LDS is involved.  The V
//       zero-points leave through  sum_t P'_t (128 + z_t)  with the SAME rounded P'.
// Everything else (split partials, last-arriver merge) is the epilogue shared with the VALU kernel.

// FUSED (round 4): the decode-step form, as span_attn_ft_mfma_kernel<.., FUSED> for the 16-bit cache.  a.q is the fused
// pre-Rotary qkv row, a.seq_lens the tokens already cached (a.len_bias = 1).  Query heads are rotated in the prologue (cos / sin
// from a.rope_tab; the rotate-half partner d +- 64 of a lane's dims lives in lane ^ 32); in the workgroup whose range holds the
// new token ONE wave rotates + quantises this step's K head and quantises its V head with the span codec (span_codec.hpp:
// byte-identical to rope_kv_append_kernel), into LDS and -- one writer per (request, group) -- from there into the span; the new
// token is then one more (single-token) block of that wave after its loop over the cached ones, built from the LDS copy.
// One launch instead of two per layer: the append kernel (896 one-wave workgroups at batch 32) cost 5 us of launch latency.
// what-if timing builds (tools/build_ksl_variant.sh NAME -DDIHIP_U4_X=.. span_attn): 1 no Rotary of q, 2 no new token, 4 loads only
// (no arithmetic in the token loop), 8 no V^T . P' half; results
// WRONG.  Batch 32 x 2048 tokens (profiles/r04v_u4_step_whatif.txt): 21.7 us per layer as first built, 20.0 without (1), 20.4
// without (2), 18.4 without both; the committed form (permlane swaps for the Rotary partner, one quantisation by the last
// wave and no barrier) takes 20.5 against 22.7 for the append launch + the op-boundary kernel.  The op-boundary kernel itself
// (profiles/r04ab_*): 19.3 us as built, 16.0 without the V^T . P' half, 14.9 with NO arithmetic in the token loop -- three
// quarters of it is launch, the length -> span pointer -> data chain, 147 KB per CU at what a CU can keep in flight, and the
// split hand-off; the nibble arithmetic that looks expensive (~300 VALU per 32 tokens) is the smaller part.
#ifndef DIHIP_U4_X
#define DIHIP_U4_X 0
#endif
template <bool FUSED>
__global__ __launch_bounds__(ATTN_THREADS) void span_attn_u4_mfma_kernel(const AttnArgs a) {
  constexpr int H = 128;
  constexpr int HC = MF_HC;
  constexpr int HB = H / 2;  // bytes per token-head row
  constexpr int NEW_WAVE = ATTN_THREADS / 64 - 1;  // FUSED: the wave that handles this step's token
  __shared__ __attribute__((aligned(16))) float lds[4 * HC * ATTN_PSTRIDE + 4];
  __shared__ __attribute__((aligned(16))) unsigned char newrow[2][FUSED ? 80 : 16];  // FUSED: {64 B nibbles, zero, scale} of the new K / V head
  unsigned* flag_lds = reinterpret_cast<unsigned*>(lds + 4 * HC * ATTN_PSTRIDE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = lane >> 4, ni = lane & 15;
  const int split = blockIdx.x;
  const int grp = blockIdx.y / a.nchunks, hc = blockIdx.y % a.nchunks;
  const int b = blockIdx.z;
  const int h0 = grp * a.hpg + hc * HC;
  const int nh = min(HC, a.hpg - hc * HC);

  const int len = (int)a.seq_lens[b] + a.len_bias;
  const int tps = ((len + a.nsplits - 1) / a.nsplits + 31) & ~31;
  const int t0 = split * tps;
  // FUSED: the split that holds this step's token (the last position) walks the CACHED tokens [t0, len - 1) in the loop and takes
  // the new one from registers at the end -- nothing inside the loop knows about it (a conditional patch of the loaded tiles
  // made hipcc drain the load queue at the join of every iteration: the whole kernel 4.6 us slower)
  const int newpos = len - 1;
  const bool has_new = FUSED && !(DIHIP_U4_X & 2) && newpos >= t0 && newpos < t0 + tps;  // workgroup-uniform (all head chunks of the group)
  int t1 = has_new ? newpos : min(len, t0 + tps);
  const void* const* ksp = a.kspans + (size_t)b * a.span_stride;
  const void* const* vsp = a.vspans + (size_t)b * a.span_stride;
  const size_t par_off = (size_t)a.g * a.S * HB;  // (zero, scale) pairs follow the data of all groups

  struct Buf {
    u32x4_t k[2];      // K rows: tile c, token c*16 + ni, bytes kb*16..
    uint32_t v[8];     // V: dword ni of token (j>>2)*16 + kb*4 + (j&3)
    f32x4_t kp[2][2];  // K {zero, scale} of tokens c*16 + kb*4 + {0,1 | 2,3}
    f32x4_t vp[2][2];
  };
  // tile bases are wave-uniform, so the span pointers are scalar loads (a vector load of the pointer would sit
  // behind the prefetched rows in the in-order vmcnt queue and drain it)
  // per-lane byte offsets inside a 16-token tile (constant over the kernel): every load below is then
  // wave-uniform 64-bit base (SGPR pair) + 32-bit lane offset (+ immediate) -- the saddr form, no per-lane
  // 64-bit address arithmetic in the loop
  const uint32_t lo_k = (uint32_t)(ni * HB + kb * 16);      // K row of token ni, bytes kb*16..
  const uint32_t lo_p = (uint32_t)(kb * 32);                // {zero, scale} x 4 tokens kb*4..
  const uint32_t lo_v = (uint32_t)(kb * 4 * HB + ni * 4);   // V dword ni of token kb*4 (+ rr*HB)
  auto issue = [&](Buf& r, int tb) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int base = tb + c * 16;
      base = base < t1 ? base : ((t1 - 1) & ~15);  // a tile past the range re-reads the last valid one (masked later)
      const int sp = __builtin_amdgcn_readfirstlane(base / a.S);
      const int pos0 = base - sp * a.S;
      const size_t row0 = (size_t)grp * a.S + pos0;  // wave-uniform
      const unsigned char* kd = reinterpret_cast<const unsigned char*>(ksp[sp]) + row0 * HB;
      const unsigned char* vd = reinterpret_cast<const unsigned char*>(vsp[sp]) + row0 * HB;
      const unsigned char* kq = reinterpret_cast<const unsigned char*>(ksp[sp]) + par_off + row0 * 8;
      const unsigned char* vq = reinterpret_cast<const unsigned char*>(vsp[sp]) + par_off + row0 * 8;
      r.k[c] = gload<u32x4_t>(kd + lo_k);
      r.kp[c][0] = gload<f32x4_t>(kq + lo_p);
      r.kp[c][1] = gload<f32x4_t>(kq + lo_p + 16);
      r.vp[c][0] = gload<f32x4_t>(vq + lo_p);
      r.vp[c][1] = gload<f32x4_t>(vq + lo_p + 16);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) r.v[c * 4 + rr] = gload<uint32_t>(vd + lo_v + rr * HB);
    }
  };

  const int tb0 = t0 + wave * MF_TOK;
  const bool active = tb0 < t1;
  Buf ba, bb;
#if defined(DIHIP_GEMV_TRACE) && DIHIP_GEMV_TRACE
  unsigned long long* const trw = a.trace ? a.trace + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + wave * 8) : nullptr;
#define DIHIP_U4_STAMP(I) do { if (trw) trw[I] = wall_clock64(); } while (0)
#else
  unsigned long long* const trw = nullptr;
#define DIHIP_U4_STAMP(I) do { } while (0)
#endif
  DIHIP_U4_STAMP(0);  // entry (lengths and span table base read)
  if (active) issue(ba, tb0);

  // ---- Q as the B operand: lane (kb, head ni) holds dims kb*32 + ks*8 + e in the order the nibble expansion
  // produces them (pairs (e, e+4)); unscaled, so the bf16 values are exact
  u32x4_t qf[4];
  float qsum = 0.f;
  const size_t qrow_stride = FUSED ? (size_t)(a.n + 2 * a.g) * H : (size_t)a.n * H;
  const float* cs_row = FUSED ? a.rope_tab + (size_t)newpos * 128 : nullptr;  // {cos, sin} of dim pairs 0 .. 63 at that position
  {
    const bool hv = ni < nh;
    const uint16_t* qrow = reinterpret_cast<const uint16_t*>(a.q) + (size_t)b * qrow_stride + (size_t)(h0 + (hv ? ni : 0)) * H + kb * 32;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qrow + ks * 8);
      if constexpr (FUSED && !(DIHIP_U4_X & 1)) {
        // Rotary (rotate-half, rope_kv_append_kernel's arithmetic): dims kb*32 + ks*8 + 2j + e; the partner d +- 64 is lane ^ 32
        u32x4_t rot;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const auto sw = __builtin_amdgcn_permlane32_swap(raw[j], raw[j], false, false);  // {lower half, upper half} in every lane: one VALU
          const uint32_t pw = kb < 2 ? sw[1] : sw[0];                                          // instruction, not a ds_bpermute round trip
          const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(cs_row + (size_t)((kb & 1) * 32 + ks * 8 + 2 * j) * 2);  // {c0, s0, c1, s1}
          float r[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float x = ft_bits_to_f32<DIHIP_BF16>(e ? raw[j] >> 16 : raw[j] & 0xFFFFu);
            const float pt = ft_bits_to_f32<DIHIP_BF16>(e ? pw >> 16 : pw & 0xFFFFu);
            r[e] = kb < 2 ? x * cs[2 * e] - pt * cs[2 * e + 1] : x * cs[2 * e] + pt * cs[2 * e + 1];
          }
          rot[j] = f32_to_ft_bits<DIHIP_BF16>(r[0]) | (f32_to_ft_bits<DIHIP_BF16>(r[1]) << 16);
        }
        raw = rot;
      }
      if (!hv) raw = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; ++j) qsum += ft_bits_to_f32<DIHIP_BF16>(raw[j] & 0xFFFFu) + ft_bits_to_f32<DIHIP_BF16>(raw[j] >> 16);
      qf[ks] = u32x4_t{(raw[0] & 0xFFFFu) | (raw[2] << 16), (raw[0] >> 16) | (raw[2] & 0xFFFF0000u),
                       (raw[1] & 0xFFFFu) | (raw[3] << 16), (raw[1] >> 16) | (raw[3] & 0xFFFF0000u)};
    }
    qsum = rows_sum(qsum);
  }

  // FUSED: this step's K (rotated, rounded, quantised) and V head of the group, as the cache will hold them
  u32x4_t knew = {};
  uint32_t vnew = 0u;
  float knz = 0.f, kns = 0.f, vnz = 0.f, vns = 0.f;
  if constexpr (FUSED) {
    if (has_new && wave == NEW_WAVE) {
      // one wave alone, K then V: lane holds d = 2 * lane, 2 * lane + 1 (the codec's element order).  No workgroup barrier: the
      // other waves start their token loops at once, and this wave reads back what it wrote itself (LDS operations of a wave
      // execute in order); its first K / V tiles are in flight meanwhile.  The LAST wave: it has the fewest 32-token blocks.
      const int sp = newpos / a.S, pos = newpos - sp * a.S;
      const bool writer = hc == 0 && sp < a.span_stride;  // one per (request, group): DecoderCacheAppend; a token past the span table is dropped, as kv_append_kernel does
#pragma unroll
      for (int kv = 0; kv < 2; ++kv) {
        const uint16_t* row = reinterpret_cast<const uint16_t*>(a.q) + (size_t)b * qrow_stride + (size_t)(a.n + (kv ? a.g : 0) + grp) * H;
        const uint32_t w = reinterpret_cast<const uint32_t*>(row)[lane];
        float x[2] = {ft_bits_to_f32<DIHIP_BF16>(w & 0xFFFFu), ft_bits_to_f32<DIHIP_BF16>(w >> 16)};
        if (kv == 0) {
          const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(cs_row + (size_t)((2 * lane) & 63) * 2);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float pt = __shfl_xor(x[e], 32, 64);
            const float r = lane < 32 ? x[e] * cs[2 * e] - pt * cs[2 * e + 1] : x[e] * cs[2 * e] + pt * cs[2 * e + 1];
            x[e] = ft_round<DIHIP_BF16>(r);
          }
        }
        store_token_head<DIHIP_BF16, DIHIP_KV_U4, 2>(newrow[kv], x, 0, 0, 1, 1, H, lane);  // a one-token "span": data, then {zero, scale}
        // (the row is read back below through pointers of other types: the byte stores must not be moved past those reads -- ADVICE r4)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (writer) {  // the same bytes into the span: 64 B of nibbles + the parameter pair
          unsigned char* span = reinterpret_cast<unsigned char*>(const_cast<void*>((kv ? vsp : ksp)[sp]));
          const size_t rowi = (size_t)grp * a.S + pos;
          if (lane < 16) gstore<uint32_t>(span + rowi * HB + lane * 4, reinterpret_cast<const uint32_t*>(newrow[kv])[lane]);
          if (lane == 16) gstore<uint64_t>(span + par_off + rowi * 8, *reinterpret_cast<const uint64_t*>(newrow[kv] + HB));
        }
      }
      knew = *reinterpret_cast<const u32x4_t*>(newrow[0] + kb * 16);
      knz = reinterpret_cast<const float*>(newrow[0] + HB)[0];
      kns = reinterpret_cast<const float*>(newrow[0] + HB)[1];
      vnew = reinterpret_cast<const uint32_t*>(newrow[1])[ni];
      vnz = reinterpret_cast<const float*>(newrow[1] + HB)[0];
      vns = reinterpret_cast<const float*>(newrow[1] + HB)[1];
    }
  }

  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f, czero = 0.f;
  f32x4_t o[8];  // O^T tile dt: rows (dims) kb*4 + r  <->  dim kb*32 + r*8 + dt, column = head ni
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) o[dt] = zero4;

  auto process = [&](const Buf& r, int tb) {
    if constexpr ((DIHIP_U4_X & 4) != 0) {  // what-if: loads only
      uint32_t x = r.k[0][0] ^ r.k[0][1] ^ r.k[0][2] ^ r.k[0][3] ^ r.k[1][0] ^ r.k[1][1] ^ r.k[1][2] ^ r.k[1][3];
#pragma unroll
      for (int j = 0; j < 8; ++j) x ^= r.v[j];
      x ^= __float_as_uint(r.kp[0][0][0] + r.kp[0][1][0] + r.kp[1][0][0] + r.kp[1][1][0] + r.vp[0][0][0] + r.vp[0][1][0] + r.vp[1][0][0] + r.vp[1][1][0]);
      l += __uint_as_float(x & 0x3FFFFFFFu);
      m = 0.f;
      return;
    }
    // ---- scores of the 32 tokens (transposed): sc[c][r] = token tb + c*16 + kb*4 + r, head ni
    float sc[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4_t acc = zero4;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint32_t w = r.k[c][ks];
        const u32x4_t kf = {(w & 0x000F000Fu) | 0x43004300u, ((w >> 4) & 0x000F000Fu) | 0x43004300u,
                            ((w >> 8) & 0x000F000Fu) | 0x43004300u, ((w >> 12) & 0x000F000Fu) | 0x43004300u};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[ks]),
                                                      acc, 0, 0, 0);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float kz = r.kp[c][rr >> 1][(rr & 1) * 2], ksc = r.kp[c][rr >> 1][(rr & 1) * 2 + 1];
        const float v = (ksc * a.scale) * fmaf(-(128.f + kz), qsum, acc[rr]);
        sc[c][rr] = tb + c * 16 + kb * 4 + rr < t1 ? v : -INFINITY;
      }
    }
    // ---- online softmax: the head's tokens of this iteration live in the 4 lanes (kb) with this ni
    float mn = m;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) mn = fmaxf(mn, sc[c][rr]);
    mn = rows_max(mn);
    const float corr = safe_exp_diff(m, mn);
    m = mn;
    float ps = 0.f;
    uint32_t pk[4], pl[4];  // P' = hi + lo, both bf16: the second MFMA pass keeps P at f32 accuracy
    float cz = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        float pv[2], zz[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int rr = h2 * 2 + e;
          const bool valid = tb + c * 16 + kb * 4 + rr < t1;
          const float p = safe_exp_diff(sc[c][rr], mn);
          ps += p;
          const float vz = r.vp[c][rr >> 1][(rr & 1) * 2], vs = r.vp[c][rr >> 1][(rr & 1) * 2 + 1];
          pv[e] = valid ? p * vs : 0.f;  // p == 0 there, but the parameters may be junk
          zz[e] = valid ? 128.f + vz : 0.f;
        }
        const uint32_t hi = pack_bf16x2(pv[0], pv[1]);
        const float h0f = __uint_as_float(hi << 16), h1f = __uint_as_float(hi & 0xFFFF0000u);
        const uint32_t lo = pack_bf16x2(pv[0] - h0f, pv[1] - h1f);
        pk[c * 2 + h2] = hi;
        pl[c * 2 + h2] = lo;
        cz = fmaf(h0f + __uint_as_float(lo << 16), zz[0], cz);
        cz = fmaf(h1f + __uint_as_float(lo & 0xFFFF0000u), zz[1], cz);
      }
    l = l * corr + ps;
    czero = czero * corr + cz;
    // the running maxima settle after the first few iterations: rescale only when some head's maximum moved
    // (wave-uniform branch), which also lets the accumulators live in the MFMA's own registers
    if (__builtin_amdgcn_ballot_w64(corr != 1.f) != 0ull) {
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) o[dt][rr] *= corr;
    }
    if constexpr ((DIHIP_U4_X & 8) != 0) {  // what-if: no V^T . P' half
      o[0][0] += __uint_as_float((pk[0] ^ pl[1] ^ r.v[0] ^ r.v[5]) & 0x3FFFFFFFu);
      return;
    }
    // ---- O^T += V^T . P': k-slot j = token (j>>2)*16 + kb*4 + (j&3) on both operands
    uint32_t le[4], lo_[4], he[4], ho[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t d0 = r.v[2 * i], d1 = r.v[2 * i + 1];
      le[i] = d0 & 0x0F0F0F0Fu;
      lo_[i] = (d0 >> 4) & 0x0F0F0F0Fu;
      he[i] = d1 & 0x0F0F0F0Fu;
      ho[i] = (d1 >> 4) & 0x0F0F0F0Fu;
    }
    const u32x4_t pkv = {pk[0], pk[1], pk[2], pk[3]};
    const u32x4_t plv = {pl[0], pl[1], pl[2], pl[3]};
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      constexpr uint32_t SEL[4] = {0x0C040C00u, 0x0C050C01u, 0x0C060C02u, 0x0C070C03u};
      u32x4_t vf;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        vf[i] = __builtin_amdgcn_perm((dt & 1) ? ho[i] : he[i], (dt & 1) ? lo_[i] : le[i], SEL[dt >> 1]) | 0x43004300u;
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pkv), o[dt], 0,
                                                      0, 0);
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, plv), o[dt], 0,
                                                      0, 0);
    }
  };

  DIHIP_U4_STAMP(1);  // first loads issued, q fragments built
  if (active) {
    constexpr int STEP = 4 * MF_TOK;  // the 4 waves interleave 32-token blocks
    for (int tb = tb0; tb < t1; tb += 2 * STEP) {
      issue(bb, tb + STEP);  // unconditional (clamped inside): see the note in the VALU kernel
      process(ba, tb);
      if (tb == tb0) DIHIP_U4_STAMP(2);  // first 32 tokens done
      issue(ba, tb + 2 * STEP);
      if (tb + STEP < t1) process(bb, tb + STEP);
    }
  }
  if constexpr (FUSED) {
    if (has_new && wave == NEW_WAVE) {
      // this step's token as a block of its own: token 0 of tile 0 is (row ni = 0, parameters / V dwords of the lanes kb = 0, rr = 0),
      // every other token of the block is masked (t1 = newpos + 1)
      Buf bn;
      bn.k[0] = bn.k[1] = knew;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          bn.kp[c][hh] = f32x4_t{knz, kns, knz, kns};
          bn.vp[c][hh] = f32x4_t{vnz, vns, vnz, vns};
        }
#pragma unroll
      for (int j = 0; j < 8; ++j) bn.v[j] = vnew;
      t1 = newpos + 1;
      process(bn, newpos);
    }
  }
  DIHIP_U4_STAMP(3);  // token loop done
  // ---- totals of the head over the 4 token rows (kb), then the record the shared epilogue expects
  l = rows_sum(l);
  czero = rows_sum(czero);
  if (ni < nh) {
    float* rec = lds + (wave * HC + ni) * ATTN_PSTRIDE;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      f32x4_t x0, x1;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        x0[dt] = o[dt][rr] - czero;
        x1[dt] = o[4 + dt][rr] - czero;
      }
      *reinterpret_cast<f32x4_t*>(rec + kb * 32 + rr * 8) = x0;
      *reinterpret_cast<f32x4_t*>(rec + kb * 32 + rr * 8 + 4) = x1;
    }
    if (kb == 0) {
      rec[H] = m;
      rec[H + 1] = l;
    }
  }
  attn_block_epilogue<DIHIP_BF16, HC>(a, lds, flag_lds, b, h0, nh, split, trw);
  DIHIP_U4_STAMP(7);
#undef DIHIP_U4_STAMP
}

// ---- 16-bit KV cache (bf16 / f16, the default cache mode) on the matrix cores -----------------------------
// Same transposed formulation as the u4 kernel (S^T = K.Q^T, lane-local softmax, O^T = V^T.P^T with P as
// bf16/f16 hi + lo).  K rows are A fragments as stored (lane (kb, token) <- 16 bytes = dims ks*32 + kb*8..).
// ------------------------------------------------------------------------------------------
struct AttnPlan {
  int HC, nchunks, nsplits;
  int waves;  // live waves per workgroup: 4, or 8 for the wide form of the 16-bit MFMA kernel (batch 1)
  bool mfma;
  size_t partial_bytes;
};

static bool attn_use_mfma(int mode, int dtype) {
  static const bool enabled = !env_off("DIHIP_ATTN_MFMA");  // =0: keep the VALU kernel for the u4 cache (diagnostics)
  if (!enabled || dtype == DIHIP_F32) return false;
  return mode == DIHIP_KV_U4 ? dtype == DIHIP_BF16 : (mode == DIHIP_KV_NONE || mode == DIHIP_KV_I8);
}

// the 16-bit cache's MFMA kernel has an 8-wave form (span_attn_ft_mfma_w8_kernel; the fused attention block runs the same body)
static bool attn_wide_ok(int mode, int dtype) {
  // opt-in (DIHIP_ATTN_WIDE=1): measured at 2048 tokens, 28 / 4 heads -- 9 split records instead of 17, but the 8-wave workgroup's own merge
  // and its two waves per SIMD in the tile phase give the gain back: block 16.4 against 15.8 us per layer (profiles/r06_attn_block_polls.txt)
  static const bool enabled = [] { const char* e = getenv("DIHIP_ATTN_WIDE"); return e && e[0] == '1'; }();
  return enabled && mode == DIHIP_KV_NONE && (dtype == DIHIP_BF16 || dtype == DIHIP_F16);
}

static AttnPlan attn_plan(int batch, int n_heads, int n_groups, int max_seq_len, int num_cus, bool mfma = false,
                          int split_cap = 256, bool wide_ok = false) {
  AttnPlan p;
  const int hpg = n_heads / n_groups;
  p.mfma = mfma;
  // batch 1, opt-in: one 8-wave workgroup per CU covers 256 tokens per pass -- half the split records of the 4-wave form
  p.waves = mfma && wide_ok && batch == 1 ? 8 : 4;
  p.HC = mfma ? MF_HC : hpg <= 1 ? 1 : hpg <= 2 ? 2 : hpg <= 4 ? 4 : 8;
  p.nchunks = (hpg + p.HC - 1) / p.HC;
  if (num_cus <= 0) num_cus = cached_num_cus();
  if (num_cus <= 0) num_cus = 256;
  const long base = (long)batch * n_groups * p.nchunks;
  static const int wgs_per_cu = std::max(0, env_int("DIHIP_ATTN_WGS_PER_CU", 0));  // workgroups (4 waves) the split count aims at per CU
  // the MFMA kernel hides latency with two co-resident workgroups per CU; the VALU kernel measured best with one
  const int per_cu = wgs_per_cu > 0 ? wgs_per_cu : (mfma && p.waves == 4 ? 2 : 1);
  long want = ((long)num_cus * per_cu + base - 1) / base;
  static const int min_tps = std::max(32, env_int("DIHIP_ATTN_SPLIT_TOKENS", 128));  // fewest tokens per split (diagnostics)
  const int split_tokens = std::max(min_tps, 32 * p.waves);                               // one pass of the workgroup
  const long max_splits = std::max(1, (max_seq_len + split_tokens - 1) / split_tokens);  // >= 128 (256) tokens per split
  p.nsplits = (int)std::max<long>(1, std::min<long>(std::min<long>(want, max_splits), split_cap));
  // two workgroups per CU only while a split keeps >= 256 tokens: below that the second workgroup buys no bandwidth and every
  // extra split is another record to merge (one TP = 8 rank of Qwen2-72B, batch 16 x 1 KV head x 4096 tokens: 16 splits 15.4 us,
  // 32 splits 18.4; the 57B MoE step at batch 16 x 4 x 1024: 4 splits = 8 + 0.4 %; profiles/r03ab)
  if (per_cu > 1 && p.nsplits * base > num_cus && (max_seq_len + p.nsplits - 1) / p.nsplits < 256) {
    const long one_per_cu = std::max<long>(1, (num_cus + base - 1) / base);
    p.nsplits = (int)std::max<long>(1, std::min<long>(p.nsplits, one_per_cu));
  }
  static const int force_splits = env_int("DIHIP_ATTN_NSPLITS", 0);  // diagnostics
  if (force_splits > 0) p.nsplits = (int)std::min<long>(std::min<long>(force_splits, max_splits), split_cap);
  p.partial_bytes = p.nsplits > 1 ? (size_t)batch * n_heads * p.nsplits * ATTN_PSTRIDE * sizeof(float) : 0;
  return p;
}

template <int FT, int MODE>
static void launch_attn(const AttnPlan& p, const AttnArgs& a, dim3 grid, hipStream_t s) {
  switch (p.HC) {
    case 1: hipLaunchKernelGGL((span_attn_decode_kernel<FT, MODE, 1>), grid, dim3(ATTN_THREADS), 0, s, a); break;
    case 2: hipLaunchKernelGGL((span_attn_decode_kernel<FT, MODE, 2>), grid, dim3(ATTN_THREADS), 0, s, a); break;
    case 4: hipLaunchKernelGGL((span_attn_decode_kernel<FT, MODE, 4>), grid, dim3(ATTN_THREADS), 0, s, a); break;
    default: hipLaunchKernelGGL((span_attn_decode_kernel<FT, MODE, 8>), grid, dim3(ATTN_THREADS), 0, s, a); break;
  }
}

// the 8-wave form of the 16-bit MFMA kernel: 72 KB of dynamic LDS (granted once per instantiation)
template <int FT, bool FUSED>
static bool launch_w8_t(dim3 grid, hipStream_t s, const AttnArgs& a) {
  constexpr int lds = ft_mfma_smem_bytes(8);
  auto kern = span_attn_ft_mfma_w8_kernel<FT, FUSED>;
  static std::atomic<bool> granted{false};
  if (!granted.load(std::memory_order_relaxed)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      set_last_error("span_attn: %d bytes of LDS refused", lds);
      return false;
    }
    granted.store(true, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
  return true;
}
static bool launch_w8(int dtype, bool fused, dim3 grid, hipStream_t s, const AttnArgs& a) {
  if (dtype == DIHIP_BF16) return fused ? launch_w8_t<DIHIP_BF16, true>(grid, s, a) : launch_w8_t<DIHIP_BF16, false>(grid, s, a);
  return fused ? launch_w8_t<DIHIP_F16, true>(grid, s, a) : launch_w8_t<DIHIP_F16, false>(grid, s, a);
}

static bool span_len_valid(int S) { return S == 16 || S == 32 || S == 64 || S == 128; }

// returns 0 or an SaStatus-like code: 3 param, 1 hip
static int run_decode(hipStream_t s, void* out, const void* q, const void* const* ks, const void* const* vs,
                      const uint32_t* seq_lens_dev, int batch, int n, int g, int H, int S, int span_stride,
                      int max_seq_len, int mode, int dtype, float scale, void* ws, size_t ws_bytes, unsigned* counters,
                      int num_cus, int out_layout = DIHIP_ACT_ROWMAJOR, int len_bias = 0, size_t counter_bytes = 0) {
  if (H != 128) {
    set_last_error("span_attn: unsupported head size %d (only 128, dispatch.hpp:45-57)", H);
    return DIHIP_SA_PARAM_ERROR;
  }
  if (n % g != 0 || n / g > 32) {
    set_last_error("span_attn: nHeads/nGroups must be an integer <= 32 (got %d/%d)", n, g);
    return DIHIP_SA_PARAM_ERROR;
  }
  if (!span_len_valid(S)) {
    set_last_error("span_attn: span length %d not in {16,32,64,128}", S);
    return DIHIP_SA_PARAM_ERROR;
  }
  const AttnPlan p = attn_plan(batch, n, g, max_seq_len, num_cus, attn_use_mfma(mode, dtype), 256, attn_wide_ok(mode, dtype));
  if (p.nsplits > 1 && (ws == nullptr || ws_bytes < p.partial_bytes)) {
    set_last_error("span_attn: workspace too small (%zu < %zu)", ws_bytes, p.partial_bytes);
    return DIHIP_SA_PARAM_ERROR;
  }
  AttnArgs a{};
  a.out = out;
  a.q = q;
  a.kspans = ks;
  a.vspans = vs;
  a.seq_lens = seq_lens_dev;
  a.len_bias = len_bias;
  a.partials = reinterpret_cast<float*>(ws);
  // Split sequences: with the caller's ticket words (`sync`: dihip_span_attn_sync_bytes, zeroed once, one 128-byte line per
  // (request, group, head chunk)) the partial records are merged INSIDE the launch -- write-through records, arrival ticket,
  // the last workgroup merges (attn_block_epilogue_wt, round 3; the fenced last-arriver form of round 1 cost ~20 us per layer
  // at batch 32 and is gone) -- else by span_attn_split_merge_kernel as a second launch.  DIHIP_ATTN_MERGE=launch: always the latter.
  static const bool merge_in_launch = [] {
    const char* e = getenv("DIHIP_ATTN_MERGE");
    return !(e && e[0] == 'l');
  }();
  //  the ticket words are only used when the caller states their size and it covers one 128-byte line per
  // (request, KV group, head chunk): the legacy entry points pass 0 and keep the two-launch merge
  const bool ticket = merge_in_launch && counters != nullptr && p.nsplits > 1 && p.partial_bytes < (1ull << 31) &&
                      counter_bytes >= (size_t)batch * g * p.nchunks * 128;
  a.counters = ticket ? counters : nullptr;
  a.merge_wt = ticket ? 1 : 0;
  a.partial_bytes = p.partial_bytes;
  a.B = batch;
  a.n = n;
  a.g = g;
  a.hpg = n / g;
  a.S = S;
  a.span_stride = span_stride;
  a.nsplits = p.nsplits;
  a.nchunks = p.nchunks;
  a.scale = scale;
  if (out_layout == DIHIP_ACT_FRAG32) {
    if (batch > 32 || dtype == DIHIP_F32) {
      set_last_error("span_attn: FRAG32 output needs batch <= 32 and 16-bit activations");
      return DIHIP_SA_PARAM_ERROR;
    }
    a.out_frag_mt = batch > 16 ? 2 : 1;
  }
  const dim3 grid(p.nsplits, g * p.nchunks, batch);
  a.trace = debug_trace_buffer((size_t)p.nsplits * g * p.nchunks * batch * 32 * sizeof(unsigned long long));
  bool ok = true;
  if (p.mfma && mode == DIHIP_KV_U4) {
    hipLaunchKernelGGL(span_attn_u4_mfma_kernel<false>, grid, dim3(ATTN_THREADS), 0, s, a);
  } else if (p.mfma && p.waves == 8) {
    if (!launch_w8(dtype, false, grid, s, a)) return DIHIP_SA_HIP_ERROR;
  } else if (p.mfma && dtype == DIHIP_BF16 && mode == DIHIP_KV_NONE) {
    hipLaunchKernelGGL((span_attn_ft_mfma_kernel<DIHIP_BF16, DIHIP_KV_NONE, false>), grid, dim3(ATTN_THREADS), 0, s, a);
  } else if (p.mfma && dtype == DIHIP_F16 && mode == DIHIP_KV_NONE) {
    hipLaunchKernelGGL((span_attn_ft_mfma_kernel<DIHIP_F16, DIHIP_KV_NONE, false>), grid, dim3(ATTN_THREADS), 0, s, a);
  } else if (p.mfma && dtype == DIHIP_BF16) {
    hipLaunchKernelGGL((span_attn_ft_mfma_kernel<DIHIP_BF16, DIHIP_KV_I8, false>), grid, dim3(ATTN_THREADS), 0, s, a);
  } else if (p.mfma) {
    hipLaunchKernelGGL((span_attn_ft_mfma_kernel<DIHIP_F16, DIHIP_KV_I8, false>), grid, dim3(ATTN_THREADS), 0, s, a);
  } else
#define GO(FTV, MODEV)                                      \
  if (dtype == FTV && mode == MODEV) {                      \
    launch_attn<FTV, MODEV>(p, a, grid, s);                 \
  } else
  GO(DIHIP_BF16, DIHIP_KV_NONE)
  GO(DIHIP_BF16, DIHIP_KV_I8)
  GO(DIHIP_BF16, DIHIP_KV_U4)
  GO(DIHIP_F16, DIHIP_KV_NONE)
  GO(DIHIP_F16, DIHIP_KV_I8)
  GO(DIHIP_F16, DIHIP_KV_U4)
  GO(DIHIP_F32, DIHIP_KV_NONE)
  GO(DIHIP_F32, DIHIP_KV_I8)
  GO(DIHIP_F32, DIHIP_KV_U4) { ok = false; }
#undef GO
  if (!ok) {
    set_last_error("span_attn: unsupported dtype %d / kv mode %d", dtype, mode);
    return DIHIP_SA_PARAM_ERROR;
  }
  if (p.nsplits > 1 && !ticket) {
    const dim3 mg(batch * n), mb(128);
    if (dtype == DIHIP_BF16)
      hipLaunchKernelGGL(span_attn_split_merge_kernel<DIHIP_BF16>, mg, mb, 0, s, out, a.partials, n, p.nsplits, a.out_frag_mt);
    else if (dtype == DIHIP_F16)
      hipLaunchKernelGGL(span_attn_split_merge_kernel<DIHIP_F16>, mg, mb, 0, s, out, a.partials, n, p.nsplits, a.out_frag_mt);
    else
      hipLaunchKernelGGL(span_attn_split_merge_kernel<DIHIP_F32>, mg, mb, 0, s, out, a.partials, n, p.nsplits, a.out_frag_mt);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error("span_attn: launch failed: %s", hipGetErrorString(e));
    return DIHIP_SA_HIP_ERROR;
  }
  return DIHIP_SA_SUCCESS;
}

// the decode-step plan of the 16-bit cache as decode_attn_block.hip runs it inside its own launch: the split count, split width
// and record buffer of span_attn_fused_mfma below (same partial records, same merge order)
void span_attn_block_plan(int batch, int n_heads, int n_groups, int max_seq_len, int* nsplits, int* nchunks, int* tps_static,
                          size_t* partial_bytes, int* waves) {
  const AttnPlan p = attn_plan(batch, n_heads, n_groups, max_seq_len, 0, true, 256, attn_wide_ok(DIHIP_KV_NONE, DIHIP_BF16));
  *nsplits = p.nsplits;
  *waves = p.waves;
  *nchunks = p.nchunks;
  *tps_static = ((max_seq_len + p.nsplits - 1) / p.nsplits + 31) & ~31;
  *partial_bytes = p.partial_bytes;
}

// decode-step form (Rotary + cache append folded in) for the 16-bit cache: span_attn_ft_mfma_kernel<FT, NONE, true>
size_t span_attn_fused_mfma_workspace_bytes(int batch, int n_heads, int n_groups, int max_seq_len) {
  return attn_plan(batch, n_heads, n_groups, max_seq_len, 0, true).partial_bytes;  // (the 4-wave plan: at least the 8-wave form's split count)
}

int span_attn_fused_mfma(void* stream, void* output, const void* qkv, void* const* k_span_array, void* const* v_span_array,
                         const uint32_t* old_seq_lens_dev, const float* rope_table, int batch, int n_heads, int n_groups,
                         int span_len, int n_spans_per_request, int max_seq_len, int kv_mode, int dtype, float qk_scale, void* ws,
                         size_t ws_bytes, bool* handled, void* sync, size_t sync_bytes, int out_layout) {
  *handled = false;
  // 16-bit and int8 cache (bf16 / f16) and uint4 cache with bf16 activations (uint4 with f16 rows keeps its append launch).
  // DIHIP_ATTN_I8_FUSED=0: the int8 cache keeps the append launch too (A/B)
  const bool u4 = kv_mode == DIHIP_KV_U4 && dtype == DIHIP_BF16;
  static const bool i8_fused = !env_off("DIHIP_ATTN_I8_FUSED");
  const bool i8 = kv_mode == DIHIP_KV_I8 && i8_fused;
  if ((kv_mode != DIHIP_KV_NONE && !u4 && !i8) || !attn_use_mfma(kv_mode, dtype)) return DIHIP_SUCCESS;
  if (out_layout == DIHIP_ACT_FRAG32 && ((!u4 && !i8) || batch > 32)) return DIHIP_SUCCESS;  // (the 16-bit form writes row-major rows)
  const AttnPlan p = attn_plan(batch, n_heads, n_groups, max_seq_len, 0, true, 256, attn_wide_ok(kv_mode, dtype));
  if (p.nsplits > 1 && (ws == nullptr || ws_bytes < p.partial_bytes)) return DIHIP_SUCCESS;  // caller's kernels size their own
  *handled = true;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  AttnArgs a{};
  a.out = output;
  a.q = qkv;
  a.kspans = k_span_array;
  a.vspans = v_span_array;
  a.seq_lens = old_seq_lens_dev;
  a.partials = reinterpret_cast<float*>(ws);
  a.B = batch;
  a.n = n_heads;
  a.g = n_groups;
  a.hpg = n_heads / n_groups;
  a.S = span_len;
  a.span_stride = n_spans_per_request;
  a.nsplits = p.nsplits;
  a.nchunks = p.nchunks;
  a.scale = qk_scale;
  a.rope_tab = rope_table;
  a.partial_bytes = p.partial_bytes;
  // split width fixed from max_seq_len (AttnArgs::tps_static); DIHIP_ATTN_STATIC_TPS=0: from the request's length (A/B)
  static const bool static_tps = !env_off("DIHIP_ATTN_STATIC_TPS");
  static const int merge_mode = [] {  // "launch": the split merge as a second launch also when a sync buffer is given (A/B)
    const char* m = getenv("DIHIP_ATTN_MERGE");
    return (m && m[0] == 'l') ? 0 : (m && m[0] == 'n') ? 2 : 1;  // "none" (timing experiments only): partials written, never merged
  }();
  if (static_tps && !u4) a.tps_static = ((max_seq_len + p.nsplits - 1) / p.nsplits + 31) & ~31;  // (the uint4 kernel splits by the request's length)
  if (u4) a.len_bias = 1;
  if (u4 || i8) a.out_frag_mt = out_layout == DIHIP_ACT_FRAG32 ? (batch > 16 ? 2 : 1) : 0;
  // in-launch merge: needs the caller's zero-initialised ticket words (dihip_span_attn_decode_fused_sync)
  const bool merge_wt = merge_mode == 1 && p.nsplits > 1 && sync != nullptr &&
                        sync_bytes >= (size_t)batch * n_groups * p.nchunks * 128 && p.partial_bytes < (1ull << 31);
  if (merge_wt) {
    a.counters = reinterpret_cast<unsigned*>(sync);
    a.merge_wt = 1;
  }
  const dim3 grid(p.nsplits, n_groups * p.nchunks, batch);
  a.trace = debug_trace_buffer((size_t)p.nsplits * n_groups * p.nchunks * batch * 32 * sizeof(unsigned long long));
  if (u4)
    hipLaunchKernelGGL(span_attn_u4_mfma_kernel<true>, grid, dim3(ATTN_THREADS), 0, s, a);
  else if (i8 && dtype == DIHIP_BF16)
    hipLaunchKernelGGL((span_attn_ft_mfma_kernel<DIHIP_BF16, DIHIP_KV_I8, true>), grid, dim3(ATTN_THREADS), 0, s, a);
  else if (i8)
    hipLaunchKernelGGL((span_attn_ft_mfma_kernel<DIHIP_F16, DIHIP_KV_I8, true>), grid, dim3(ATTN_THREADS), 0, s, a);
  else if (p.waves == 8) {
    if (!launch_w8(dtype, true, grid, s, a)) return DIHIP_RUNTIME_ERROR;
  } else if (dtype == DIHIP_BF16)
    hipLaunchKernelGGL((span_attn_ft_mfma_kernel<DIHIP_BF16, DIHIP_KV_NONE, true>), grid, dim3(ATTN_THREADS), 0, s, a);
  else
    hipLaunchKernelGGL((span_attn_ft_mfma_kernel<DIHIP_F16, DIHIP_KV_NONE, true>), grid, dim3(ATTN_THREADS), 0, s, a);
  if (p.nsplits > 1 && !merge_wt && merge_mode != 2) {
    const dim3 mg(batch * n_heads), mb(128);
    if (dtype == DIHIP_BF16)
      hipLaunchKernelGGL(span_attn_split_merge_kernel<DIHIP_BF16>, mg, mb, 0, s, output, a.partials, n_heads, p.nsplits, a.out_frag_mt);
    else
      hipLaunchKernelGGL(span_attn_split_merge_kernel<DIHIP_F16>, mg, mb, 0, s, output, a.partials, n_heads, p.nsplits, 0);
  }
  return launch_status();
}

}  // namespace dihip

using namespace dihip;

extern "C" {

int dihip_span_attn_merge_partials(void* stream, void* output, const float* partials, int batch, int n_heads, int nsplits,
                                   int dtype) {
  DIHIP_REQUIRE(output && partials && batch >= 0 && n_heads > 0 && nsplits >= 1, DIHIP_PARAM_ERROR,
                "span_attn_merge_partials: invalid parameter");
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16 || dtype == DIHIP_F32, DIHIP_PARAM_ERROR, "span_attn_merge_partials: dtype");
  if (batch == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 mg(batch * n_heads), mb(128);
  if (dtype == DIHIP_BF16)
    hipLaunchKernelGGL(span_attn_split_merge_kernel<DIHIP_BF16>, mg, mb, 0, s, output, partials, n_heads, nsplits, 0);
  else if (dtype == DIHIP_F16)
    hipLaunchKernelGGL(span_attn_split_merge_kernel<DIHIP_F16>, mg, mb, 0, s, output, partials, n_heads, nsplits, 0);
  else
    hipLaunchKernelGGL(span_attn_split_merge_kernel<DIHIP_F32>, mg, mb, 0, s, output, partials, n_heads, nsplits, 0);
  return launch_status();
}

}  // extern "C"

// handle of the reference-shaped API (span::SpanAttnHandle, span-attention/src/attn/span_attn_handle.hpp)
struct dihip_span_attn_handle {
  int dtype, kv_mode, batch, n_heads, n_groups, head_size, span_len, n_spans, num_cus, max_len;
  std::vector<uint32_t> seq_lens;
  size_t lens_bytes, counter_bytes, partial_bytes;
};

extern "C" {

size_t dihip_span_attn_sync_bytes(int batch, int n_heads) {
  if (batch <= 0 || n_heads <= 0) return 0;
  // one 128-byte line per (request, KV group, head chunk) -- at most one per head -- for the in-launch merge's arrival words
  return (size_t)batch * n_heads * 128;
}

size_t dihip_span_attn_decode_workspace_bytes(int batch, int n_heads, int head_size, int max_seq_len, int num_cus) {
  if (batch <= 0 || n_heads <= 0 || max_seq_len <= 0) return 0;
  (void)head_size;
  // worst case over the group count: g = 1 .. n gives at most 256 splits
  size_t worst = 0;
  for (int g = 1; g <= n_heads; ++g) {
    if (n_heads % g) continue;
    worst = std::max(worst, attn_plan(batch, n_heads, g, max_seq_len, num_cus, false).partial_bytes);
    worst = std::max(worst, attn_plan(batch, n_heads, g, max_seq_len, num_cus, true).partial_bytes);
  }
  return worst + 256;
}

}  // extern "C"

namespace dihip {
int span_attn_decode_biased(void* stream, void* output, const void* query, const void* const* k_span_array,
                            const void* const* v_span_array, const uint32_t* seq_lens_dev, int len_bias, int batch, int n_heads,
                            int n_groups, int span_len, int n_spans_per_request, int max_seq_len, int kv_mode, int dtype,
                            float qk_scale, void* ws, size_t ws_bytes) {
  const int st = run_decode(reinterpret_cast<hipStream_t>(stream), output, query, k_span_array, v_span_array, seq_lens_dev, batch,
                            n_heads, n_groups, 128, span_len, n_spans_per_request, max_seq_len, kv_mode, dtype, qk_scale, ws,
                            ws_bytes, nullptr, 0, DIHIP_ACT_ROWMAJOR, len_bias);
  if (st == DIHIP_SA_SUCCESS) return DIHIP_SUCCESS;
  return st == DIHIP_SA_PARAM_ERROR ? DIHIP_PARAM_ERROR : DIHIP_RUNTIME_ERROR;
}
size_t span_attn_decode_workspace_bytes(int batch, int n_heads, int n_groups, int max_seq_len, int kv_mode, int dtype) {
  return attn_plan(batch, n_heads, n_groups, max_seq_len, 0, attn_use_mfma(kv_mode, dtype)).partial_bytes;  // (4-wave plan: an upper bound)
}
}  // namespace dihip

extern "C" {

int dihip_span_attn_decode(void* stream, void* output, const void* query, const void* const* k_span_array,
                           const void* const* v_span_array, const uint32_t* seq_lens_dev, int batch, int n_heads,
                           int n_groups, int head_size, int span_len, int n_spans_per_request, int max_seq_len,
                           int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes, void* sync) {
  return dihip_span_attn_decode_ex(stream, output, query, k_span_array, v_span_array, seq_lens_dev, batch, n_heads, n_groups,
                                   head_size, span_len, n_spans_per_request, max_seq_len, kv_mode, dtype, qk_scale, ws,
                                   ws_bytes, sync, DIHIP_ACT_ROWMAJOR);
}

int dihip_span_attn_decode_ex(void* stream, void* output, const void* query, const void* const* k_span_array,
                              const void* const* v_span_array, const uint32_t* seq_lens_dev, int batch, int n_heads,
                              int n_groups, int head_size, int span_len, int n_spans_per_request, int max_seq_len,
                              int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes, void* sync,
                              int out_layout) {
  (void)sync;  // legacy signature: never touched (two-launch merge), as the header promises
  return dihip_span_attn_decode_sync(stream, output, query, k_span_array, v_span_array, seq_lens_dev, batch, n_heads, n_groups,
                                     head_size, span_len, n_spans_per_request, max_seq_len, kv_mode, dtype, qk_scale, ws, ws_bytes,
                                     nullptr, 0, out_layout);
}

int dihip_span_attn_decode_sync(void* stream, void* output, const void* query, const void* const* k_span_array,
                                const void* const* v_span_array, const uint32_t* seq_lens_dev, int batch, int n_heads,
                                int n_groups, int head_size, int span_len, int n_spans_per_request, int max_seq_len,
                                int kv_mode, int dtype, float qk_scale, void* ws, size_t ws_bytes, void* sync,
                                size_t sync_bytes, int out_layout) {
  DIHIP_REQUIRE(out_layout == DIHIP_ACT_ROWMAJOR || out_layout == DIHIP_ACT_FRAG32, DIHIP_PARAM_ERROR,
                "span_attn_decode: bad out_layout");
  DIHIP_REQUIRE(batch >= 0 && n_heads > 0 && n_groups > 0 && head_size > 0 && n_spans_per_request > 0 &&
                    max_seq_len > 0,
                DIHIP_PARAM_ERROR, "span_attn_decode: invalid parameter");
  DIHIP_REQUIRE(output && query && k_span_array && v_span_array && seq_lens_dev, DIHIP_PARAM_ERROR,
                "span_attn_decode: null pointer");
  if (batch == 0) return DIHIP_SUCCESS;
  int st = run_decode(reinterpret_cast<hipStream_t>(stream), output, query, k_span_array, v_span_array, seq_lens_dev,
                      batch, n_heads, n_groups, head_size, span_len, n_spans_per_request, max_seq_len, kv_mode, dtype,
                      qk_scale, ws, ws_bytes, reinterpret_cast<unsigned*>(sync), 0, out_layout, 0, sync ? sync_bytes : 0);
  if (st == DIHIP_SA_SUCCESS) return DIHIP_SUCCESS;
  return st == DIHIP_SA_PARAM_ERROR ? DIHIP_PARAM_ERROR : DIHIP_RUNTIME_ERROR;
}

// ---- reference-shaped handle API (span_attn.h:108-175, api.cpp:41-177) --------------------------
int dihip_span_attn_create_handle(dihip_span_attn_handle_t* handle, int dtype, int kv_mode, int batch, int n_heads,
                                  int n_groups, int head_size, int span_len, int n_spans_per_request,
                                  const int* seq_len_host, int num_cus) {
  if (batch <= 0 || n_heads <= 0 || n_groups <= 0 || head_size <= 0 || span_len <= 0 || n_spans_per_request <= 0) {
    set_last_error("CreateHandle: invalid parameter");
    return DIHIP_SA_PARAM_ERROR;
  }
  if (n_heads % n_groups != 0) {
    set_last_error("CreateHandle: nHeads should be a multiple of nGroups");
    return DIHIP_SA_PARAM_ERROR;
  }
  if (seq_len_host == nullptr || handle == nullptr) {
    set_last_error("CreateHandle: null pointer");
    return DIHIP_SA_PARAM_ERROR;
  }
  if (dtype != DIHIP_F32 && dtype != DIHIP_F16 && dtype != DIHIP_BF16) return DIHIP_SA_PARAM_ERROR;
  if (kv_mode != DIHIP_KV_NONE && kv_mode != DIHIP_KV_I8 && kv_mode != DIHIP_KV_U4) return DIHIP_SA_PARAM_ERROR;
  if (head_size != 128) {  // span_attention.hpp:75-84
    set_last_error("unsupported head size: %d", head_size);
    return DIHIP_SA_PARAM_ERROR;
  }
  int max_len = 0;
  for (int i = 0; i < batch; ++i) {
    if (seq_len_host[i] < 0) return DIHIP_SA_PARAM_ERROR;
    max_len = std::max(max_len, seq_len_host[i]);
  }
  // the reference caps a request at 65535 tiles of 64 tokens (tile_mapping.hpp:42-49)
  if ((long)(max_len + 63) / 64 > 65535) {
    set_last_error("CreateHandle: sequence length %d exceeds the tile limit", max_len);
    return DIHIP_SA_EXCEED_LIMIT_ERROR;
  }
  dihip_span_attn_handle* h = new (std::nothrow) dihip_span_attn_handle();
  if (!h) return DIHIP_SA_RUNTIME_ERROR;
  h->dtype = dtype;
  h->kv_mode = kv_mode;
  h->batch = batch;
  h->n_heads = n_heads;
  h->n_groups = n_groups;
  h->head_size = head_size;
  h->span_len = span_len;
  h->n_spans = n_spans_per_request;
  h->num_cus = num_cus;
  h->max_len = std::max(1, max_len);
  h->seq_lens.assign(seq_len_host, seq_len_host + batch);
  h->lens_bytes = ((size_t)batch * sizeof(uint32_t) + 255) & ~(size_t)255;
  h->counter_bytes = dihip_span_attn_sync_bytes(batch, n_heads);
  h->partial_bytes = attn_plan(batch, n_heads, n_groups, h->max_len, num_cus, attn_use_mfma(kv_mode, dtype)).partial_bytes;
  *handle = h;
  return DIHIP_SA_SUCCESS;
}

int dihip_span_attn_destroy_handle(dihip_span_attn_handle_t handle) {
  if (handle == nullptr) return DIHIP_SA_PARAM_ERROR;
  delete handle;
  return DIHIP_SA_SUCCESS;
}

int dihip_span_attn_host_workspace_bytes(size_t* bytes, dihip_span_attn_handle_t handle) {
  if (handle == nullptr || bytes == nullptr) return DIHIP_SA_PARAM_ERROR;
  *bytes = handle->lens_bytes;  // staging of the sequence lengths for the async H2D copy
  return DIHIP_SA_SUCCESS;
}

int dihip_span_attn_device_workspace_bytes(size_t* bytes, dihip_span_attn_handle_t handle) {
  if (handle == nullptr || bytes == nullptr) return DIHIP_SA_PARAM_ERROR;
  *bytes = handle->lens_bytes + handle->counter_bytes + handle->partial_bytes + 256;
  return DIHIP_SA_SUCCESS;
}

int dihip_span_attn_run(void* output, const void* query, const void* const* k_span_array,
                        const void* const* v_span_array, void* device_ws, size_t device_ws_bytes, void* host_ws,
                        size_t host_ws_bytes, float qk_scale, dihip_span_attn_handle_t handle, void* stream) {
  if (handle == nullptr) return DIHIP_SA_PARAM_ERROR;
  if (output == nullptr || query == nullptr || k_span_array == nullptr || v_span_array == nullptr) {
    set_last_error("Run: input and output pointers must not be null");
    return DIHIP_SA_PARAM_ERROR;
  }
  if ((device_ws == nullptr && device_ws_bytes > 0) || (host_ws == nullptr && host_ws_bytes > 0)) {
    set_last_error("Run: workspace pointer must not be null if its size is non-zero");
    return DIHIP_SA_PARAM_ERROR;
  }
  // parameter checks that the reference defers to Run (dispatch.hpp:45-81)
  if (handle->n_heads / handle->n_groups > 32 || !span_len_valid(handle->span_len)) {
    set_last_error("Run: unsupported heads-per-group %d or span length %d", handle->n_heads / handle->n_groups,
                   handle->span_len);
    return DIHIP_SA_PARAM_ERROR;
  }
  size_t need = 0;
  dihip_span_attn_device_workspace_bytes(&need, handle);
  if (device_ws == nullptr || device_ws_bytes < need || host_ws == nullptr || host_ws_bytes < handle->lens_bytes) {
    set_last_error("Run: workspace too small");
    return DIHIP_SA_PARAM_ERROR;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  char* dws = reinterpret_cast<char*>(device_ws);
  uint32_t* lens_dev = reinterpret_cast<uint32_t*>(dws);
  unsigned* counters = reinterpret_cast<unsigned*>(dws + handle->lens_bytes);
  void* partials = dws + handle->lens_bytes + handle->counter_bytes;
  std::copy(handle->seq_lens.begin(), handle->seq_lens.end(), reinterpret_cast<uint32_t*>(host_ws));
  if (hipMemcpyAsync(lens_dev, host_ws, (size_t)handle->batch * sizeof(uint32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemsetAsync(counters, 0, handle->counter_bytes, s) != hipSuccess) {
    (void)hipGetLastError();
    return DIHIP_SA_HIP_ERROR;
  }
  return run_decode(s, output, query, k_span_array, v_span_array, lens_dev, handle->batch, handle->n_heads,
                    handle->n_groups, handle->head_size, handle->span_len, handle->n_spans, handle->max_len,
                    handle->kv_mode, handle->dtype, qk_scale, partials, handle->partial_bytes, counters,
                    handle->num_cus, DIHIP_ACT_ROWMAJOR, 0, handle->counter_bytes);
}

int dihip_debug_attn_plan(int batch, int n_heads, int n_groups, int max_seq_len, int kv_mode, int dtype, int num_cus, int* nsplits,
                          int* mfma) {
  if (batch <= 0 || n_heads <= 0 || n_groups <= 0 || n_heads % n_groups || max_seq_len <= 0) return DIHIP_SA_PARAM_ERROR;
  const bool m = attn_use_mfma(kv_mode, dtype);
  const AttnPlan p = attn_plan(batch, n_heads, n_groups, max_seq_len, num_cus, m, 256, attn_wide_ok(kv_mode, dtype));
  if (nsplits) *nsplits = p.nsplits;
  if (mfma) *mfma = m ? 1 : 0;
  return DIHIP_SUCCESS;
}

}  // extern "C"
