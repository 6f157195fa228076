// span_attn_fused.hip -- decode-step attention with Rotary and the KV-cache append folded in
// (include/dashinfer_hip.h section 3b).
//
// The reference runs Rotary -> DecOptMQA (cache append kernel, then QK / softmax / PV / reduce
// kernels) per layer (python/pyhie/allspark/model/qwen_v15.py:228-262, span_attn_op.cpp:90-169).
// At batch 1 every one of those launches is latency, not bandwidth.  This file does the whole
// sub-graph in two launches that read the fused qkv row the GEMV just produced:
//
//   span_attn_fused_kernel  grid (kv-splits, kv-groups x head-chunks, requests).  A workgroup
//     rotates its query heads in registers (cos/sin from a table built once with the same
//     arithmetic as dihip_rope_qk), streams its token range of the request's K/V spans exactly
//     once (16 lanes x 16 B per token-head row, every row reused by all heads of the GQA group) and
//     leaves one (max, sum, o[128]) partial per head.  The workgroup whose range contains the new
//     token also rotates / rounds / quantises this step's K and V head, writes them into the span
//     (byte-identical to DecoderCacheAppend) and uses the values the cache now holds.
//   span_attn_merge_kernel  one workgroup per (request, head): combines the split partials.
//     A separate launch is cheaper here than an in-kernel last-arriver hand-off (one kernel
//     boundary ~1.5 us vs release fence + ticket + acquire ~4-6 us, MI355X_MICROARCH.md price list).
//
// No inter-workgroup communication, no host work per step: sequence lengths are read on the device.
#include <algorithm>
#include <cstdlib>

#include "span_attn_common.hpp"

namespace dihip {

struct FusedArgs {
  const void* qkv;  // FT [B, (n + 2g) * H], pre-Rotary (bias already applied by the GEMV)
  void* const* kspans;
  void* const* vspans;
  const uint32_t* old_lens;  // tokens already cached per request (= position of the new token)
  const float* rope_tab;     // [max_pos][64] {cos, sin}
  float* partials;           // [B][n][nsplits][ATTN_PSTRIDE]
  int B, n, g, hpg, S, span_stride, nsplits, nchunks, tps;
  float scale;
  unsigned long long* trace;  // diagnostics (dihip_debug_set_trace): [workgroup][4 waves][8] stamps, or null
};

__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true)));
  return v;
}
__device__ __forceinline__ float row16_min(float v) { return -row16_max(-v); }

// 8 dims (d = dc*8 ..) of one head row of the fused qkv tensor, optionally rotated (rotate-half,
// csrc/core/kernel/cpu/rotary.cpp:22-106) and rounded to FT like the Rotary op's output tensor.
template <int FT, bool ROPE>
__device__ __forceinline__ void load_head_row(float (&x)[8], const void* qkv, size_t off, int dc, const float* cs_row) {
  static_assert(FT != DIHIP_F32, "16-bit activations");
  const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(qkv) + off + dc * 8);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x[2 * j] = ft_bits_to_f32<FT>(raw[j] & 0xFFFFu);
    x[2 * j + 1] = ft_bits_to_f32<FT>(raw[j] >> 16);
  }
  if constexpr (ROPE) {
    // partner dims d +- 64 live 8 lanes away in the 16-lane row
    const f32x4_t* t = reinterpret_cast<const f32x4_t*>(cs_row + (dc & 7) * 16);
    const f32x4_t c0 = t[0], c1 = t[1], c2 = t[2], c3 = t[3];  // {cos,sin} x 8 dims
    const float cosv[8] = {c0[0], c0[2], c1[0], c1[2], c2[0], c2[2], c3[0], c3[2]};
    const float sinv[8] = {c0[1], c0[3], c1[1], c1[3], c2[1], c2[3], c3[1], c3[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float partner =
          __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[j]), 0x128, 0xF, 0xF, true));
      const float r = dc < 8 ? x[j] * cosv[j] - partner * sinv[j] : x[j] * cosv[j] + partner * sinv[j];
      x[j] = ft_round<FT>(r);
    }
  }
}

// Quantise (cache mode MODE) + store the row x (16 lanes x 8 dims of one token-head) exactly like
// store_token_head (span_cache.hip), and replace x by what a reader of the cache will decode.
template <int FT, int MODE>
__device__ __forceinline__ void append_row(float (&x)[8], void* span, int head, int pos, int g, int S, int dc,
                                           bool do_store) {
  constexpr int H = 128;
  if constexpr (MODE == DIHIP_KV_NONE) {
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = f32_to_ft_bits<FT>(x[2 * j]) | (f32_to_ft_bits<FT>(x[2 * j + 1]) << 16);
    if (do_store) *reinterpret_cast<u32x4_t*>(reinterpret_cast<uint16_t*>(span) + ((size_t)head * S + pos) * H + dc * 8) = o;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = ft_round<FT>(x[j]);
  } else {
    constexpr float QMAX = MODE == DIHIP_KV_I8 ? 127.f : 15.f;
    constexpr float QMIN = MODE == DIHIP_KV_I8 ? -128.f : 0.f;
    float mx = x[0], mn = x[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      mx = fmaxf(mx, x[j]);
      mn = fminf(mn, x[j]);
    }
    mx = row16_max(mx);
    mn = row16_min(mn);
    float qs = (mx - mn) / (QMAX - QMIN);
    qs = fmaxf(qs, 1e-5f);
    float qz = QMIN - mn / qs;
    qz = fminf(qz, QMAX);
    if constexpr (MODE == DIHIP_KV_I8) qz = fmaxf(qz, QMIN);
    qz = rintf(qz);
    constexpr int HB = MODE == DIHIP_KV_I8 ? H : H / 2;
    unsigned char* base = reinterpret_cast<unsigned char*>(span);
    unsigned char* data = base + ((size_t)head * S + pos) * HB;
    int q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = fminf(qz + x[j] / qs, QMAX);
      if constexpr (MODE == DIHIP_KV_I8) t = fmaxf(t, QMIN);
      t = rintf(t);
      if constexpr (MODE == DIHIP_KV_U4) t = fmaxf(t, 0.f);  // saturating float -> u32 (impl_u4.cuh:79-93)
      q[j] = (int)t;
    }
    if constexpr (MODE == DIHIP_KV_I8) {
      u32x2_t o = {0u, 0u};
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j >> 2] |= ((uint32_t)q[j] & 0xFFu) << (8 * (j & 3));
      if (do_store) *reinterpret_cast<u32x2_t*>(data + dc * 8) = o;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = ((float)(int)(signed char)(q[j] & 0xFF) - qz) * qs;
    } else {
      uint32_t o = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) o |= ((uint32_t)q[j] & 0xFu) << (4 * j);
      if (do_store) *reinterpret_cast<uint32_t*>(data + dc * 4) = o;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = ((float)(q[j] & 0xF) - qz) * qs;
    }
    if (do_store && dc == 0) {
      float* params = reinterpret_cast<float*>(base + (size_t)g * S * HB) + ((size_t)head * S + pos) * 2;
      params[0] = qz;
      params[1] = qs;
    }
  }
}

// Workgroup = (kv split, kv group, request); wave w owns query heads [w*HPW, (w+1)*HPW) of the group
// and walks ALL tokens of the split: every wave reads the same K/V rows (L1/L2 hits after the
// first), which costs a few redundant cache reads but removes every cross-wave reduction, keeps
// the per-wave instruction stream short (one or two heads: rope, scores, online softmax, P.V,
// slot merge) and needs no barrier except around the new token.
#ifndef DIHIP_FUSED_GL
#define DIHIP_FUSED_GL 1
#endif
constexpr bool FUSED_GLOBAL_LOADS = DIHIP_FUSED_GL != 0;
constexpr int FUSED_TB = 8;             // tokens per lane slot per iteration (4 slots x 8 = 32 tokens)
constexpr int FUSED_TOK_PER_ITER = 32;
constexpr int FUSED_MAX_WAVES = 8;

template <int FT, int MODE, int HPW>
__global__ __launch_bounds__(64 * FUSED_MAX_WAVES) void span_attn_fused_kernel(const FusedArgs a) {
  constexpr int H = 128;
  constexpr int TB = FUSED_TB;
  constexpr int HC = HPW;
  __shared__ __attribute__((aligned(16))) float lds[2 * H];
  float* knew = lds;  // this step's K / V head as the cache holds them
  float* vnew = knew + H;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = lane >> 4, dc = lane & 15;
  const int split = blockIdx.x;
  const int grp = blockIdx.y;
  const int b = blockIdx.z;
  const int h0 = grp * a.hpg + wave * HC;
  const int nh = max(0, min(HC, a.hpg - wave * HC));

#define DIHIP_ATTN_STAMP(I)                                                                       \
  do {                                                                                            \
    if (a.trace && lane == 0)                                                                     \
      a.trace[((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * FUSED_MAX_WAVES + wave) * 8 + (I)] = wall_clock64(); \
  } while (0)
  DIHIP_ATTN_STAMP(0);
  // The sequence length and the span pointers of the split's first 32-token block (one or, for 16-token spans,
  // two spans) are independent scalar loads issued together: one scalar-cache round trip instead of a chain of
  // three vector loads (length -> pointer -> rows), and nothing of it sits in the vmcnt queue.
  const void* const* ksp = a.kspans + (size_t)b * a.span_stride;
  const void* const* vsp = a.vspans + (size_t)b * a.span_stride;
  const int t0 = split * a.tps;
  const int sp0 = t0 / a.S, sp1 = min((t0 + 16) / a.S, a.span_stride - 1);
  uint32_t newpos_u;
  const void *kp0, *kp1, *vp0, *vp1;
  asm volatile(
      "s_load_dword %0, %5, 0x0\n\ts_load_dwordx2 %1, %6, 0x0\n\ts_load_dwordx2 %2, %7, 0x0\n\t"
      "s_load_dwordx2 %3, %8, 0x0\n\ts_load_dwordx2 %4, %9, 0x0\n\ts_waitcnt lgkmcnt(0)"
      : "=&s"(newpos_u), "=&s"(kp0), "=&s"(kp1), "=&s"(vp0), "=&s"(vp1)
      : "s"(a.old_lens + b), "s"(ksp + sp0), "s"(ksp + sp1), "s"(vsp + sp0), "s"(vsp + sp1)
      : "memory");
  const int newpos = (int)newpos_u;
  const int len = newpos + 1;
  const int t1 = min(len, t0 + a.tps);
  if (t0 >= t1) {
    // dead split (beyond the sequence): leave neutral partials so that the merge kernel needs no length
    if (nh > 0 && tl == 0) {
#pragma unroll
      for (int h = 0; h < HC; ++h) {
        if (h < nh) {
          float* rec = a.partials + (((size_t)b * a.n + h0 + h) * a.nsplits + split) * ATTN_PSTRIDE;
          *reinterpret_cast<f32x4_t*>(rec + dc * 8) = f32x4_t{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4_t*>(rec + dc * 8 + 4) = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (dc == 0) {
            rec[H] = -INFINITY;
            rec[H + 1] = 0.f;
          }
        }
      }
    }
    return;
  }
  const bool has_new = newpos >= t0;  // newpos < t1 always; only the last live split holds it

  auto issue = [&](KvChunk<FT, MODE> (&kc)[TB], KvChunk<FT, MODE> (&vc)[TB], int tb) {
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int t = tb + i * 4 + tl;
      const int tt = t < t1 ? t : t0;  // clamp: keeps the address legal, result discarded
      const int sp = tt / a.S, pos = tt - sp * a.S;
      kv_issue<FT, MODE, FUSED_GLOBAL_LOADS>(kc[i], ksp[sp], grp, pos, a.g, a.S, dc);
      kv_issue<FT, MODE, FUSED_GLOBAL_LOADS>(vc[i], vsp[sp], grp, pos, a.g, a.S, dc);
    }
  };
  KvChunk<FT, MODE> k0[TB], v0[TB], k1[TB], v1[TB];
  // first rows in flight before anything else (the slot of the new token reads whatever the span holds; it is
  // overridden below); rows past the range re-read the split's first row
#pragma unroll
  for (int i = 0; i < TB; ++i) {
    const int t = t0 + i * 4 + tl;
    const bool live = t < t1;
    const int pos = (live ? t : t0) & (a.S - 1);  // span lengths are powers of two
    const bool second = live && i * 4 >= 16;
    kv_issue<FT, MODE, FUSED_GLOBAL_LOADS>(k0[i], second ? kp1 : kp0, grp, pos, a.g, a.S, dc);
    kv_issue<FT, MODE, FUSED_GLOBAL_LOADS>(v0[i], second ? vp1 : vp0, grp, pos, a.g, a.S, dc);
  }
  DIHIP_ATTN_STAMP(1);

  const size_t row = (size_t)b * (a.n + 2 * a.g) * H;
  const float* cs_row = a.rope_tab + (size_t)newpos * 128;
  // rotated, FT-rounded, pre-scaled query rows of this wave's heads
  float qr[HC][8];
#pragma unroll
  for (int h = 0; h < HC; ++h) {
    if (h < nh) {
      load_head_row<FT, true>(qr[h], a.qkv, row + (size_t)(h0 + h) * H, dc, cs_row);
#pragma unroll
      for (int j = 0; j < 8; ++j) qr[h][j] *= a.scale;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) qr[h][j] = 0.f;
    }
  }
  DIHIP_ATTN_STAMP(2);
  if (has_new) {  // workgroup-uniform
    // the last two waves (least loaded when hpg is not a multiple of HPW) prepare the new token:
    // K (rotate, round, quantise, store) and V; row 0 of the wave = 16 lanes x 8 dims
    const int nw = (int)(blockDim.x >> 6);
    const int wk_ = nw - 1, wv_ = nw > 1 ? nw - 2 : 0;
    const int sp = newpos / a.S, pos = newpos - sp * a.S;
    if (wave == wk_ && tl == 0) {
      float x[8];
      load_head_row<FT, true>(x, a.qkv, row + (size_t)(a.n + grp) * H, dc, cs_row);
      append_row<FT, MODE>(x, const_cast<void*>(ksp[sp]), grp, pos, a.g, a.S, dc, true);
#pragma unroll
      for (int j = 0; j < 8; ++j) knew[dc * 8 + j] = x[j];
    }
    if (wave == wv_ && tl == (nw > 1 ? 0 : 1)) {
      float x[8];
      load_head_row<FT, false>(x, a.qkv, row + (size_t)(a.n + a.g + grp) * H, dc, cs_row);
      append_row<FT, MODE>(x, const_cast<void*>(vsp[sp]), grp, pos, a.g, a.S, dc, true);
#pragma unroll
      for (int j = 0; j < 8; ++j) vnew[dc * 8 + j] = x[j];
    }
    __syncthreads();
  }
  DIHIP_ATTN_STAMP(3);

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
    float vx[TB][8];
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int t = tb + i * 4 + tl;
      const bool valid = t < t1;
      float kx[8];
      kv_decode<FT, MODE>(kc[i], kx);
      kv_decode<FT, MODE>(vc[i], vx[i]);
      if (has_new && t == newpos) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          kx[j] = knew[dc * 8 + j];
          vx[i][j] = vnew[dc * 8 + j];
        }
      }
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
        s[i][h] = safe_exp_diff(s[i][h], mn);
        ps += s[i][h];
      }
      l[h] = l[h] * corr + ps;
      m[h] = mn;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[h][j] *= corr;
    }
#pragma unroll
    for (int i = 0; i < TB; ++i)
#pragma unroll
      for (int h = 0; h < HC; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[h][j] = fmaf(s[i][h], vx[i][j], o[h][j]);
  };

  if (nh > 0) {  // wave-uniform (idle waves of a ragged last head chunk only help with the new token)
    constexpr int STEP = FUSED_TOK_PER_ITER;
    // (conditional prefetch: the unconditional form that keeps hipcc's vmcnt counting exact across iterations
    // measured 2 us slower at the batch-1 plan of one iteration per split)
    for (int tb = t0; tb < t1; tb += 2 * STEP) {
      const bool more1 = tb + STEP < t1;
      if (more1) issue(k1, v1, tb + STEP);
      process(k0, v0, tb);
      if (more1) {
        if (tb + 2 * STEP < t1) issue(k0, v0, tb + 2 * STEP);
        process(k1, v1, tb + STEP);
      }
    }
    DIHIP_ATTN_STAMP(4);

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
    DIHIP_ATTN_STAMP(5);
    if (tl == 0) {
#pragma unroll
      for (int h = 0; h < HC; ++h) {
        if (h < nh) {
          float* rec = a.partials + (((size_t)b * a.n + h0 + h) * a.nsplits + split) * ATTN_PSTRIDE;
          *reinterpret_cast<f32x4_t*>(rec + dc * 8) = f32x4_t{o[h][0], o[h][1], o[h][2], o[h][3]};
          *reinterpret_cast<f32x4_t*>(rec + dc * 8 + 4) = f32x4_t{o[h][4], o[h][5], o[h][6], o[h][7]};
          if (dc == 0) {
            rec[H] = m[h];
            rec[H + 1] = l[h];
          }
        }
      }
    }
  }
  DIHIP_ATTN_STAMP(6);
#undef DIHIP_ATTN_STAMP
}

// one workgroup (128 threads = head dims) per (request, head).  All partial loads of a batch of 16
// splits are issued together: max, weights and the weighted sum of the batch cost one round trip.
template <int FT>
__global__ __launch_bounds__(128) void span_attn_merge_kernel(void* out, const float* partials, const uint32_t* old_lens,
                                                              int n, int nsplits, int tps) {
  constexpr int H = 128;
  const int bh = blockIdx.x, d = threadIdx.x;
  (void)old_lens;
  (void)n;
  (void)tps;
  const int ns = nsplits;  // dead splits hold neutral records (max = -inf, sum = 0, o = 0)
  const float* base = partials + (size_t)bh * nsplits * ATTN_PSTRIDE;
  float mm = -INFINITY, ll = 0.f, oo = 0.f;
  constexpr int MB = 72;  // splits per batch: all loads of a batch are in flight together
  for (int sb = 0; sb < ns; sb += MB) {
    float mv[MB], lv[MB], ov[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const bool in = sb + j < ns;
      const float* rec = base + (size_t)(in ? sb + j : 0) * ATTN_PSTRIDE;
      mv[j] = rec[H];
      lv[j] = rec[H + 1];
      ov[j] = rec[d];
      if (!in) {
        mv[j] = -INFINITY;
        lv[j] = 0.f;
        ov[j] = 0.f;
      }
    }
    float bm = mm;
#pragma unroll
    for (int j = 0; j < MB; ++j) bm = fmaxf(bm, mv[j]);
    const float carry = safe_exp_diff(mm, bm);
    ll *= carry;
    oo *= carry;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      const float c = safe_exp_diff(mv[j], bm);
      ll = fmaf(lv[j], c, ll);
      oo = fmaf(ov[j], c, oo);
    }
    mm = bm;
  }
  store_ft<FT>(out, (size_t)bh * H + d, ll > 0.f ? oo / ll : 0.f);
}

__global__ void rope_table_kernel(float* tab, const float* inv_freq, int max_pos, int half) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= max_pos * half) return;
  const int pos = idx / half, i = idx - pos * half;
  float sn, cs;
  rope_sincos((uint32_t)pos, inv_freq[i], &sn, &cs);
  tab[(size_t)idx * 2] = cs;
  tab[(size_t)idx * 2 + 1] = sn;
}

struct FusedPlan {
  int HPW, nwaves, nsplits, tps;
  size_t partial_bytes;
};

static FusedPlan fused_plan(int batch, int n_heads, int n_groups, int max_seq_len, int num_cus) {
  FusedPlan p;
  const int hpg = n_heads / n_groups;
  p.HPW = hpg <= 8 ? 1 : hpg <= 16 ? 2 : 4;
  p.nwaves = (hpg + p.HPW - 1) / p.HPW;
  if (num_cus <= 0) num_cus = cached_num_cus();
  if (num_cus <= 0) num_cus = 256;
  const long base = (long)batch * n_groups;
  const long want = std::max<long>(1, (num_cus + base - 1) / base);  // about one workgroup per CU
  static int min_tps = -1;  // at least this many tokens per split: every extra split costs merge work
  if (min_tps < 0) {
    const char* e = getenv("DIHIP_ATTN_MIN_TPS");
    min_tps = e ? atoi(e) : 32;
    if (min_tps < FUSED_TOK_PER_ITER) min_tps = FUSED_TOK_PER_ITER;
  }
  const long max_splits = std::max(1, (max_seq_len + min_tps - 1) / min_tps);
  const int ns = (int)std::min<long>(std::min<long>(want, max_splits), 256);
  p.tps = ((max_seq_len + ns - 1) / ns + FUSED_TOK_PER_ITER - 1) / FUSED_TOK_PER_ITER * FUSED_TOK_PER_ITER;
  p.nsplits = (max_seq_len + p.tps - 1) / p.tps;
  p.partial_bytes = (size_t)batch * n_heads * p.nsplits * ATTN_PSTRIDE * sizeof(float);
  return p;
}

template <int FT, int MODE>
static void launch_fused(const FusedPlan& p, const FusedArgs& a, dim3 grid, hipStream_t s) {
  const dim3 block(64 * p.nwaves);
  switch (p.HPW) {
    case 1: hipLaunchKernelGGL((span_attn_fused_kernel<FT, MODE, 1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((span_attn_fused_kernel<FT, MODE, 2>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((span_attn_fused_kernel<FT, MODE, 4>), grid, block, 0, s, a); break;
  }
}

}  // namespace dihip

using namespace dihip;

extern "C" {

int dihip_rope_table(void* stream, float* table, const float* inv_freq, int max_pos, int head_size) {
  DIHIP_REQUIRE(table && inv_freq && max_pos > 0 && head_size > 0 && head_size % 2 == 0, DIHIP_PARAM_ERROR,
                "rope_table: bad argument");
  const int half = head_size / 2;
  const int total = max_pos * half;
  hipLaunchKernelGGL(rope_table_kernel, dim3((total + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), table,
                     inv_freq, max_pos, half);
  return launch_status();
}

size_t dihip_span_attn_fused_workspace_bytes(int batch, int n_heads, int n_groups, int head_size, int max_seq_len) {
  if (batch <= 0 || n_heads <= 0 || n_groups <= 0 || max_seq_len <= 0 || n_heads % n_groups) return 0;
  (void)head_size;
  return std::max(fused_plan(batch, n_heads, n_groups, max_seq_len, 0).partial_bytes,
                  span_attn_fused_mfma_workspace_bytes(batch, n_heads, n_groups, max_seq_len)) + 256;
}

int dihip_span_attn_decode_fused(void* stream, void* output, const void* qkv, void* const* k_span_array,
                                 void* const* v_span_array, const uint32_t* old_seq_lens_dev, const float* rope_table,
                                 int batch, int n_heads, int n_groups, int head_size, int span_len,
                                 int n_spans_per_request, int max_seq_len, int kv_mode, int dtype, float qk_scale, void* ws,
                                 size_t ws_bytes) {
  DIHIP_REQUIRE(batch >= 0 && n_heads > 0 && n_groups > 0 && n_spans_per_request > 0 && max_seq_len > 0, DIHIP_PARAM_ERROR,
                "span_attn_decode_fused: invalid parameter");
  DIHIP_REQUIRE(output && qkv && k_span_array && v_span_array && old_seq_lens_dev && rope_table, DIHIP_PARAM_ERROR,
                "span_attn_decode_fused: null pointer");
  DIHIP_REQUIRE(head_size == 128, DIHIP_PARAM_ERROR, "span_attn: unsupported head size %d (only 128, dispatch.hpp:45-57)",
                head_size);
  DIHIP_REQUIRE(n_heads % n_groups == 0 && n_heads / n_groups <= 32, DIHIP_PARAM_ERROR,
                "span_attn: nHeads/nGroups must be an integer <= 32 (got %d/%d)", n_heads, n_groups);
  DIHIP_REQUIRE(span_len == 16 || span_len == 32 || span_len == 64 || span_len == 128, DIHIP_PARAM_ERROR,
                "span_attn: span length %d not in {16,32,64,128}", span_len);
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "span_attn_decode_fused: 16-bit activations only");
  DIHIP_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0, DIHIP_PARAM_ERROR, "span_attn_decode_fused: qkv must be 16-byte aligned");
  if (batch == 0) return DIHIP_SUCCESS;
  {
    // 16-bit cache: both contractions on the matrix cores (span_attn.hip); 10.7 vs 15.5 us per layer at batch 1
    bool handled = false;
    const int st = span_attn_fused_mfma(stream, output, qkv, k_span_array, v_span_array, old_seq_lens_dev, rope_table, batch,
                                        n_heads, n_groups, span_len, n_spans_per_request, max_seq_len, kv_mode, dtype, qk_scale,
                                        ws, ws_bytes, &handled);
    if (handled) return st;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const FusedPlan p = fused_plan(batch, n_heads, n_groups, max_seq_len, 0);
  DIHIP_REQUIRE(ws && ws_bytes >= p.partial_bytes, DIHIP_MEMORY_ERROR, "span_attn_decode_fused: workspace too small (%zu < %zu)",
                ws_bytes, p.partial_bytes);
  FusedArgs a{};
  a.qkv = qkv;
  a.kspans = k_span_array;
  a.vspans = v_span_array;
  a.old_lens = old_seq_lens_dev;
  a.rope_tab = rope_table;
  a.partials = reinterpret_cast<float*>(ws);
  a.B = batch;
  a.n = n_heads;
  a.g = n_groups;
  a.hpg = n_heads / n_groups;
  a.S = span_len;
  a.span_stride = n_spans_per_request;
  a.nsplits = p.nsplits;
  a.nchunks = 1;
  a.tps = p.tps;
  a.scale = qk_scale;
  const dim3 grid(p.nsplits, n_groups, batch);
  a.trace = debug_trace_buffer((size_t)p.nsplits * n_groups * batch * FUSED_MAX_WAVES * 64);
  static int dbg_phase = -1;  // diagnostics: DIHIP_ATTN_PHASE=1 main kernel only, =2 merge only
  if (dbg_phase < 0) {
    const char* e = getenv("DIHIP_ATTN_PHASE");
    dbg_phase = e ? atoi(e) : 0;
  }
  bool ok = true;
  if (dbg_phase != 2) {
#define GO(FTV, MODEV)                         \
  if (dtype == FTV && kv_mode == MODEV) {      \
    launch_fused<FTV, MODEV>(p, a, grid, s);   \
  } else
  GO(DIHIP_BF16, DIHIP_KV_NONE)
  GO(DIHIP_BF16, DIHIP_KV_I8)
  GO(DIHIP_BF16, DIHIP_KV_U4)
  GO(DIHIP_F16, DIHIP_KV_NONE)
  GO(DIHIP_F16, DIHIP_KV_I8)
  GO(DIHIP_F16, DIHIP_KV_U4) { ok = false; }
#undef GO
  }
  DIHIP_REQUIRE(ok, DIHIP_PARAM_ERROR, "span_attn_decode_fused: unsupported dtype %d / kv mode %d", dtype, kv_mode);
  if (dbg_phase == 1) return launch_status();
  if (dtype == DIHIP_BF16)
    hipLaunchKernelGGL(span_attn_merge_kernel<DIHIP_BF16>, dim3(batch * n_heads), dim3(128), 0, s, output, a.partials,
                       old_seq_lens_dev, n_heads, p.nsplits, p.tps);
  else
    hipLaunchKernelGGL(span_attn_merge_kernel<DIHIP_F16>, dim3(batch * n_heads), dim3(128), 0, s, output, a.partials,
                       old_seq_lens_dev, n_heads, p.nsplits, p.tps);
  return launch_status();
}

}  // extern "C"
