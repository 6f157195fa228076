// span_attn_ft_mfma.hpp -- the MFMA decode attention over 16-bit / int8 caches (plus the block epilogues it shares with the
// other decode kernels of span_attn.hip).
#pragma once
#include "span_attn_common.hpp"
#include "span_codec.hpp"

namespace dihip {

// Block epilogue shared by the decode kernels: the 4 waves have left one (o[128], m, l) record per head in
// `lds` ([wave][HC] records of ATTN_PSTRIDE floats).  Combines them and writes the output, or, for split
// sequences, the block's partial record for span_attn_split_merge_kernel.
// NW = live waves of the workgroup (4; 8 for the wide form of the 16-bit MFMA kernel: span_attn_ft_mfma_body).
template <int FT, int HC, bool GRAN = false, int NW = 4>
__device__ __forceinline__ void attn_block_epilogue_wt(const AttnArgs& a, float* lds, unsigned* flag_lds, int b, int h0, int nh,
                                                       int split, unsigned* counter, unsigned long long* tr,
                                                       const AttnHandoff* ho = nullptr, int grp = 0);

template <int FT, int HC, int NW = 4>
__device__ __forceinline__ void attn_block_epilogue(const AttnArgs& a, float* lds, unsigned* flag_lds, int b, int h0, int nh,
                                                    int split, unsigned long long* tr = nullptr) {
  constexpr int H = 128;
  const int tid = threadIdx.x;
  if (a.merge_wt) {  // in-launch merge of the split partials (ticket words from the caller): see attn_block_epilogue_wt
    attn_block_epilogue_wt<FT, HC, false, NW>(a, lds, flag_lds, b, h0, nh, split, a.counters + ((size_t)b * gridDim.y + blockIdx.y) * 32, tr);
    return;
  }
  __syncthreads();
  // thread -> (head, dim) pairs of the block result; kept in registers for the epilogue
  constexpr int NT = NW * 64;
  constexpr int PER_THREAD = (HC * H + NT - 1) / NT;
  float bo[PER_THREAD], bm[PER_THREAD], bl[PER_THREAD];
#pragma unroll
  for (int e = 0; e < PER_THREAD; ++e) {
    const int idx = tid + e * NT;
    const int h = idx / H, d = idx - h * H;
    bo[e] = 0.f;
    bm[e] = -INFINITY;
    bl[e] = 0.f;
    if (h < nh) {
      float mm = -INFINITY;
#pragma unroll
      for (int w = 0; w < NW; ++w) mm = fmaxf(mm, lds[(w * HC + h) * ATTN_PSTRIDE + H]);
      float ll = 0.f, oo = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float* rec = lds + (w * HC + h) * ATTN_PSTRIDE;
        const float c = safe_exp_diff(rec[H], mm);
        ll += rec[H + 1] * c;
        oo += rec[d] * c;
      }
      bo[e] = oo;
      bm[e] = mm;
      bl[e] = ll;
    }
  }

  if (a.nsplits > 1 || a.force_partials) {
#pragma unroll
    for (int e = 0; e < PER_THREAD; ++e) {
      const int idx = tid + e * NT;
      const int h = idx / H, d = idx - h * H;
      if (h < nh) {
        float* rec = a.partials + (((size_t)b * a.n + h0 + h) * a.nsplits + split) * ATTN_PSTRIDE;
        rec[d] = bo[e];
        if (d == 0) {
          rec[H] = bm[e];
          rec[H + 1] = bl[e];
        }
      }
    }
    return;  // merged by span_attn_split_merge_kernel (next launch)
  }
#pragma unroll
  for (int e = 0; e < PER_THREAD; ++e) {
    const int idx = tid + e * NT;
    const int h = idx / H, d = idx - h * H;
    if (h < nh) {
      const size_t idx = a.out_frag_mt ? act_frag_index(b, (h0 + h) * H + d, a.out_frag_mt) : ((size_t)b * a.n + h0 + h) * H + d;
      store_ft<FT>(a.out, idx, bl[e] > 0.f ? bo[e] / bl[e] : 0.f);
    }
  }
}

// the last-arriving workgroup's part of attn_block_epilogue_wt: all nsplits records of its heads, read past the L1 (sc1),
// merged in split order, output written.  MB = records per load batch.
// GRAN (fused attention block): the merged output leaves as granules for the o-projection's workgroups instead of FT rows
template <int FT, int MB, bool GRAN = false, int NW = 4, typename RSRC>
__device__ __forceinline__ void merge_split_records(const AttnArgs& a, RSRC rsrc, int b, int h0, int nh, unsigned long long* tr,
                                                    const AttnHandoff* ho = nullptr) {
  constexpr int H = 128;
  const int tid = threadIdx.x;
  for (int e = tid; e < nh * 32; e += NW * 64) {
    const int h = e >> 5, dq = e & 31;
    const uint32_t hoff = (uint32_t)(((size_t)b * a.n + h0 + h) * a.nsplits * ATTN_PSTRIDE * sizeof(float));
    // merge_order4 (span_attn_common.hpp): the maximum of ALL splits first (plans beyond one load batch: a pass over the {m, l} words) ...
    float M = -INFINITY;
    if (a.nsplits > MB) {
      for (int sb = 0; sb < a.nsplits; sb += MB) {
        u32x2_t mv[MB];
#pragma unroll
        for (int j = 0; j < MB; ++j)
          mv[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, hoff + (uint32_t)(min(sb + j, a.nsplits - 1) * (ATTN_PSTRIDE * (int)sizeof(float))) + H * 4, 0, 16);
#pragma unroll
        for (int j = 0; j < MB; ++j) M = fmaxf(M, __uint_as_float(mv[j][0]));  // (a repeated last record changes no maximum)
      }
    }
    // ... then four fma chains over the splits j = r (mod 4) in ascending order, combined as (s0 + s1) + (s2 + s3)
    float l4[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4_t o4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    static_assert(MB % 4 == 0, "a load batch keeps the chains' phase");
    for (int sb = 0; sb < a.nsplits; sb += MB) {
      u32x4_t ov[MB];
      u32x2_t mv[MB];
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        const uint32_t ro = hoff + (uint32_t)(min(sb + j, a.nsplits - 1) * (ATTN_PSTRIDE * (int)sizeof(float)));
        ov[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ro + dq * 16, 0, 16 /* sc1 */);
        mv[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ro + H * 4, 0, 16);
      }
      if (a.nsplits <= MB) {
#pragma unroll
        for (int j = 0; j < MB; ++j) M = fmaxf(M, __uint_as_float(mv[j][0]));
      }
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        if (sb + j < a.nsplits) {
          const float c = safe_exp_diff(__uint_as_float(mv[j][0]), M);
          l4[j & 3] = fmaf(__uint_as_float(mv[j][1]), c, l4[j & 3]);
#pragma unroll
          for (int q = 0; q < 4; ++q) o4[j & 3][q] = fmaf(__uint_as_float(ov[j][q]), c, o4[j & 3][q]);
        }
      }
    }
    const float ll = (l4[0] + l4[1]) + (l4[2] + l4[3]);
    f32x4_t oo;
#pragma unroll
    for (int q = 0; q < 4; ++q) oo[q] = (o4[0][q] + o4[1][q]) + (o4[2][q] + o4[3][q]);
#if defined(DIHIP_GEMV_TRACE) && DIHIP_GEMV_TRACE
    if (tr) tr[6] = wall_clock64();
#endif
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = ll > 0.f ? oo[q] / ll : 0.f;
    if constexpr (GRAN) {
      // two granules (4 dims) in one 16-byte write-through store: each granule is whole inside its aligned 8-byte half
      const auto grsrc = __builtin_amdgcn_make_buffer_rsrc(ho->out_gran, 0, (int)ho->out_gran_bytes, 0x00020000);
      const u32x4_t gv = {pack_ft2<FT>(r[0], r[1]), ho->tag, pack_ft2<FT>(r[2], r[3]), ho->tag};
      __builtin_amdgcn_raw_buffer_store_b128(gv, grsrc, (uint32_t)((((size_t)b * a.n + h0 + h) * (H / 2) + dq * 2) * 8), 0, 16 /* sc1 */);
      continue;
    }
    if constexpr (FT != DIHIP_F32) {
      if (!a.out_frag_mt) {  // row-major output: the lane's 4 dims are one 8-byte store (same round-to-nearest-even bits as store_ft)
        const u32x2_t pk = {pack_ft2<FT>(r[0], r[1]), pack_ft2<FT>(r[2], r[3])};
        *reinterpret_cast<u32x2_t*>(reinterpret_cast<uint16_t*>(a.out) + ((size_t)b * a.n + h0 + h) * H + dq * 4) = pk;
        continue;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int d = dq * 4 + q;
      const size_t idx = a.out_frag_mt ? act_frag_index(b, (h0 + h) * H + d, a.out_frag_mt) : ((size_t)b * a.n + h0 + h) * H + d;
      store_ft<FT>(a.out, idx, r[q]);
    }
  }
}

// In-launch merge of the split partials without fences (AttnArgs::merge_wt; MI355X guide, Guideline 16 form R1): the 4 waves
// have left their (o[128], m, l) records in `lds`.  The block record goes to `partials` with 16-byte WRITE-THROUGH stores
// (sc1: the line leaves this XCD's L2), every wave drains its stores, one lane takes an arrival ticket (relaxed agent-scope
// atomic), and the workgroup that arrives last for its (request, group, head chunk) reads all nsplits records with sc1
// loads -- which are served past this CU's L1, and the writers' lines are not in any other L2 -- merges them in split order
// (the arithmetic of span_attn_split_merge_kernel) and writes the FT output.  The ticket word is left at zero for the
// next launch.  Thread -> (head h = e / 32, dims (e % 32) * 4 .. + 3).
// (Measured alternatives, profiles/r03_attn_timeline.txt: "designated mergers" -- the workgroup of split s waits for all
// arrivals and merges head s, so that a group's heads merge on 7 CUs in parallel -- shorten the merge reads from 2.6 to 1.2 us
// but see the last arrival 1.4 us late through their poll: 10.6 vs 10.2 us per layer, removed; the hand-off as a whole
// costs ~3 us (store drain ~1, ticket ~0.5, reads of freshly handed-off records ~1.5-2.5) either way.)
template <int FT, int HC, bool GRAN, int NW>
__device__ __forceinline__ void attn_block_epilogue_wt(const AttnArgs& a, float* lds, unsigned* flag_lds, int b, int h0, int nh,
                                                       int split, unsigned* counter, unsigned long long* tr,
                                                       const AttnHandoff* ho, int grp) {
  constexpr int H = 128;
  const int tid = threadIdx.x;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.partials, 0, (int)a.partial_bytes, 0x00020000);
  __syncthreads();
  for (int e = tid; e < nh * 32; e += NW * 64) {
    const int h = e >> 5, dq = e & 31;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) mm = fmaxf(mm, lds[(w * HC + h) * ATTN_PSTRIDE + H]);
    float ll = 0.f;
    f32x4_t oo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* rec = lds + (w * HC + h) * ATTN_PSTRIDE;
      const float c = safe_exp_diff(rec[H], mm);
      ll += rec[H + 1] * c;
      const f32x4_t ov = *reinterpret_cast<const f32x4_t*>(rec + dq * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) oo[j] += ov[j] * c;
    }
    const uint32_t roff = (uint32_t)((((size_t)b * a.n + h0 + h) * a.nsplits + split) * ATTN_PSTRIDE * sizeof(float));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, oo), rsrc, roff + dq * 16, 0, 16 /* sc1 */);
    if (dq == 0) {
      const u32x2_t ml = {__float_as_uint(mm), __float_as_uint(ll)};
      __builtin_amdgcn_raw_buffer_store_b64(ml, rsrc, roff + H * 4, 0, 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores are acknowledged
#if defined(DIHIP_GEMV_TRACE) && DIHIP_GEMV_TRACE
  if (tr) tr[4] = wall_clock64();
#endif
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == (unsigned)a.nsplits - 1u;
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // launches of a stream do not overlap
    *flag_lds = last ? 1u : 0u;
  }
  __syncthreads();
#if defined(DIHIP_GEMV_TRACE) && DIHIP_GEMV_TRACE
  if (tr) tr[5] = wall_clock64();
#endif
  if (*flag_lds == 0u) return;
  // records per batch: all loads of a batch are in flight together.  The batch width follows the split count (8 / 12 / 16 / 20 / 24),
  // so that a 17-split plan issues 20 record loads per lane, not 32 (the excess re-reads the last record)
  if (a.nsplits <= 8) merge_split_records<FT, 8, GRAN, NW>(a, rsrc, b, h0, nh, tr, ho);
  else if (a.nsplits <= 12) merge_split_records<FT, 12, GRAN, NW>(a, rsrc, b, h0, nh, tr, ho);
  else if (a.nsplits <= 16) merge_split_records<FT, 16, GRAN, NW>(a, rsrc, b, h0, nh, tr, ho);
  else if (a.nsplits <= 20) merge_split_records<FT, 20, GRAN, NW>(a, rsrc, b, h0, nh, tr, ho);  // (17 splits of a 2048-token history: every record
                                                                                                //  beyond the count is two more loads per lane: round 6)
  else if (a.nsplits <= 24) merge_split_records<FT, 24, GRAN, NW>(a, rsrc, b, h0, nh, tr, ho);
  else merge_split_records<FT, 16, GRAN, NW>(a, rsrc, b, h0, nh, tr, ho);  // (beyond 24: the maximum pass, then batches of 16 -- four chains of
                                                                           //  accumulators beside a 32-wide batch do not fit the registers)
  // (GRAN: no flag behind the granules -- the consumers poll the group's first granule, then sweep)
}

// Polled split records (AttnHandoff::rec; fused attention block, round 6).  Replaces write-through + drain + ticket + reload of
// attn_block_epilogue_wt on the block's critical path (profiles/r06_attn_block_timeline.txt: tiles done -> merged granules 4.8 us,
// 2.9 of them one workgroup pulling 17 x 7 records = 123 KB per pass through its CU's memory queue):
//   * every split workgroup stores its record COMPLEMENTED (~bits) into the launch's record buffer (zero beforehand) with 16-byte
//     write-through stores -- no drain, no ticket;
//   * the merge is DISTRIBUTED: the group's (head, 4-dim) items are dealt over its split workgroups (14 each at 17 splits x 7
//     heads); the owner of an item polls the item's chunk of every split record until none is zero (the data is the flag: a 16-byte
//     store lands whole; ~6 KB per pass and workgroup), merges with the arithmetic of every split merge of this library (merge_order4,
//     span_attn_common.hpp: bit-identical output) and publishes the item's output granules;
//   * measured and dropped (profiles/r06_attn_block_polls.txt): all four waves polling a slice out of step (+1.4 us: a poll pass is 34
//     memory-side loads per lane, and four times the passes delay the stores they wait for), sleeping before the first poll (no effect);
//   * two record buffers alternate by launch parity: the owner zeroes its chunks of the OTHER buffer (the previous launch's records,
//     whose readers are long gone) for the next launch -- a reader never races a reset.
// FOUR lanes per item (round 6, second half): lane r of an item's quad polls the records of the splits j = r, r + 4, r + 8, ... only --
// ceil(nsplits / 4) records per lane and pass instead of all of them (17 splits: 10 loads instead of 34; a poll pass is what the detection of
// the last record is quantised in) -- and merges them; the quads combine with two DPP steps.  The arithmetic is merge_order4
// (span_attn_common.hpp): lane r's chain IS chain r, the DPP steps are (s0 + s1) + (s2 + s3) in every lane (float addition commutes bit for bit).
template <int FT, int MB4, int NW>
__device__ __forceinline__ void merge_polled_items4(const AttnArgs& a, const AttnHandoff* ho, int b, int h0, int i0, int cnt,
                                                    unsigned long long* tr) {
  constexpr int H = 128;
  constexpr uint32_t RB = ATTN_PSTRIDE * (uint32_t)sizeof(float);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 3;
  const uint32_t half = ho->rec_bytes;  // bytes of one buffer
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ho->rec, 0, (int)(2 * half), 0x00020000);
  const auto grsrc = __builtin_amdgcn_make_buffer_rsrc(ho->out_gran, 0, (int)ho->out_gran_bytes, 0x00020000);
  const uint32_t cur = ho->parity ? half : 0u, oth = ho->parity ? 0u : half;
  for (int e0 = (tid >> 6) * 16; e0 < cnt; e0 += NW * 16) {  // whole waves, 16 items each (the retry is wave-uniform)
    const int item = i0 + min(e0 + (lane >> 2), cnt - 1);
    const int h = item >> 5, dq = item & 31;
    const uint32_t hoff = (uint32_t)(((size_t)b * a.n + h0 + h) * a.nsplits) * RB;
    u32x4_t ov[MB4];
    u32x2_t mv[MB4];
    for (unsigned spins = 0;; ++spins) {
#pragma unroll
      for (int j = 0; j < MB4; ++j) {
        const uint32_t ro = cur + hoff + (uint32_t)min(r + 4 * j, a.nsplits - 1) * RB;
        ov[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ro + dq * 16, 0, 16 /* sc1 */);
        mv[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, ro + H * 4, 0, 16);
      }
      bool ok = true;
#pragma unroll
      for (int j = 0; j < MB4; ++j) ok = ok && ((ov[j][0] | ov[j][1] | ov[j][2] | ov[j][3]) != 0u) && ((mv[j][0] | mv[j][1]) != 0u);
      if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
      if (spins > ho->spin_limit) {  // gives up (wave-uniform): garbage results, flagged, never a hang
        if (lane == 0) __hip_atomic_store(ho->err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
#ifndef DIHIP_AB_SLEEP_M
#define DIHIP_AB_SLEEP_M 2
#endif
      __builtin_amdgcn_s_sleep(DIHIP_AB_SLEEP_M);
    }
#if defined(DIHIP_GEMV_TRACE) && DIHIP_GEMV_TRACE
    if (tr) tr[6] = wall_clock64();
#endif
    float bm = -INFINITY;
#pragma unroll
    for (int j = 0; j < MB4; ++j)
      if (r + 4 * j < a.nsplits) bm = fmaxf(bm, __uint_as_float(~mv[j][0]));
    bm = fmaxf(bm, dpp_mov_f32<0xB1>(bm));  // quad_perm [1,0,3,2]
    bm = fmaxf(bm, dpp_mov_f32<0x4E>(bm));  // quad_perm [2,3,0,1]
    float ll = 0.f;
    f32x4_t oo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MB4; ++j) {
      if (r + 4 * j < a.nsplits) {
        const float c = safe_exp_diff(__uint_as_float(~mv[j][0]), bm);
        ll = fmaf(__uint_as_float(~mv[j][1]), c, ll);
#pragma unroll
        for (int q = 0; q < 4; ++q) oo[q] = fmaf(__uint_as_float(~ov[j][q]), c, oo[q]);
      }
    }
    ll += dpp_mov_f32<0xB1>(ll);
    ll += dpp_mov_f32<0x4E>(ll);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      oo[q] += dpp_mov_f32<0xB1>(oo[q]);
      oo[q] += dpp_mov_f32<0x4E>(oo[q]);
    }
    float rr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rr[q] = ll > 0.f ? oo[q] / ll : 0.f;
    if (e0 + (lane >> 2) < cnt) {
      if (r == 0) {
        const u32x4_t gv = {pack_ft2<FT>(rr[0], rr[1]), ho->tag, pack_ft2<FT>(rr[2], rr[3]), ho->tag};
        __builtin_amdgcn_raw_buffer_store_b128(gv, grsrc, (uint32_t)((((size_t)b * a.n + h0 + h) * (H / 2) + dq * 2) * 8), 0, 16 /* sc1 */);
      }
      // this item's chunks of the OTHER buffer back to zero for the next launch (write-through: no line stays in this XCD's L2): lane r its splits
      const u32x4_t z4 = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < MB4; ++j) {
        if (r + 4 * j < a.nsplits) {
          const uint32_t ro = oth + hoff + (uint32_t)(r + 4 * j) * RB;
          __builtin_amdgcn_raw_buffer_store_b128(z4, rsrc, ro + dq * 16, 0, 16);
          if (dq == 0) __builtin_amdgcn_raw_buffer_store_b128(z4, rsrc, ro + H * 4, 0, 16);
        }
      }
    }
  }
}

template <int FT, int HC, int NW>
__device__ __forceinline__ void attn_block_epilogue_polled(const AttnArgs& a, float* lds, int b, int h0, int nh, int split,
                                                           unsigned long long* tr, const AttnHandoff* ho) {
  constexpr int H = 128;
  const int tid = threadIdx.x;
  const uint32_t half = ho->rec_bytes;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ho->rec, 0, (int)(2 * half), 0x00020000);
  const uint32_t cur = ho->parity ? half : 0u;
  __syncthreads();
  for (int e = tid; e < nh * 32; e += NW * 64) {  // the block record: the arithmetic of attn_block_epilogue_wt
    const int h = e >> 5, dq = e & 31;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) mm = fmaxf(mm, lds[(w * HC + h) * ATTN_PSTRIDE + H]);
    float ll = 0.f;
    f32x4_t oo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* rec = lds + (w * HC + h) * ATTN_PSTRIDE;
      const float c = safe_exp_diff(rec[H], mm);
      ll += rec[H + 1] * c;
      const f32x4_t ov = *reinterpret_cast<const f32x4_t*>(rec + dq * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) oo[j] += ov[j] * c;
    }
    const uint32_t roff = cur + (uint32_t)((((size_t)b * a.n + h0 + h) * a.nsplits + split) * ATTN_PSTRIDE * sizeof(float));
    const u32x4_t ob = __builtin_bit_cast(u32x4_t, oo);
    const u32x4_t oc = {~ob[0], ~ob[1], ~ob[2], ~ob[3]};
    __builtin_amdgcn_raw_buffer_store_b128(oc, rsrc, roff + dq * 16, 0, 16 /* sc1 */);
    if (dq == 0) {
      const u32x4_t ml = {~__float_as_uint(mm), ~__float_as_uint(ll), ~0u, ~0u};
      __builtin_amdgcn_raw_buffer_store_b128(ml, rsrc, roff + H * 4, 0, 16);
    }
  }
#if defined(DIHIP_GEMV_TRACE) && DIHIP_GEMV_TRACE
  if (tr) tr[4] = wall_clock64();
#endif
  // this workgroup's slice of the group's items (no drain: the stores are on their way; the owners' polls see them land)
  const int items = nh * 32, per = (items + a.nsplits - 1) / a.nsplits;
  const int i0 = split * per, cnt = min(per, items - i0);
  if (cnt <= 0) return;
  // a quad of lanes per item, ceil(nsplits / 4) records per lane and pass (merge_polled_items4)
  switch ((a.nsplits + 3) >> 2) {  // records per lane: the quad's lanes share the splits (nsplits <= 32: the host's contract)
    case 1: merge_polled_items4<FT, 1, NW>(a, ho, b, h0, i0, cnt, tr); break;
    case 2: merge_polled_items4<FT, 2, NW>(a, ho, b, h0, i0, cnt, tr); break;
    case 3: merge_polled_items4<FT, 3, NW>(a, ho, b, h0, i0, cnt, tr); break;
    case 4: merge_polled_items4<FT, 4, NW>(a, ho, b, h0, i0, cnt, tr); break;
    case 5: merge_polled_items4<FT, 5, NW>(a, ho, b, h0, i0, cnt, tr); break;
    case 6: merge_polled_items4<FT, 6, NW>(a, ho, b, h0, i0, cnt, tr); break;
    case 7: merge_polled_items4<FT, 7, NW>(a, ho, b, h0, i0, cnt, tr); break;
    default: merge_polled_items4<FT, 8, NW>(a, ho, b, h0, i0, cnt, tr); break;
  }
}

constexpr int MF_HC = 16;   // query heads per workgroup chunk (MFMA N)
constexpr int MF_TOK = 32;  // tokens per wave iteration

// V^T needs 8 tokens of one dim per lane: the 32-token V tile goes through a per-wave LDS tile (coalesced
// 16-byte loads, ds_write_b128, row pitch 288 B) and comes back with ds_read_b64_tr_b16, the gfx950 transpose
// read (tools/trread_test.cpp pins its lane mapping): two reads give the 8 k-slots of one dim tile, pitch 288
// makes the 16 lanes of a read hit 32 distinct banks.  Registers hold ONE K tile pair and ONE V tile: V(t+1) is
// requested as soon as V(t) sits in LDS, K(t+1) as soon as the scores of t are done -- the loads are in
// flight during softmax and P.V without a second register set.
constexpr int MF_VPITCH = 288;  // bytes per token row of the LDS V tile (256 + 32)

template <int FT>
__device__ __forceinline__ f32x4_t mfma_ft(const u32x4_t& a_, const u32x4_t& b_, const f32x4_t& c_) {
  if constexpr (FT == DIHIP_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a_), __builtin_bit_cast(bf16x8_t, b_), c_, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a_), __builtin_bit_cast(f16x8_t, b_), c_, 0, 0, 0);
}


// MODE = DIHIP_KV_NONE: rows are FT.  MODE = DIHIP_KV_I8: rows are int8 with per-token {zero, scale}; bytes become
// exact FT integers 128 + q (byte ^ 0x80 -> v_cvt_f32_ubyte -> packed convert) on the way to the K fragments / the LDS
// V tile, and zero-points and scales are applied to the f32 scores and to P exactly as in the u4 kernel.
// FUSED (16-bit cache only): the decode-step form.  a.q is the fused pre-Rotary qkv row, a.seq_lens the tokens already
// cached.  Query heads are rotated in the prologue (the rotate-half partner d +- 64 of a lane's dims is k-step ks +- 2 of
// the SAME lane); the workgroup whose range holds the new token rotates / rounds this step's K head and takes its V
// head, one wave writes both into the span (byte-identical to DecoderCacheAppend), and every lane whose (clamped)
// token is the new one uses the register copy -- the span row itself may not be written yet.
// NW (16-bit cache only): live waves per workgroup.  4: two workgroups share a CU.  8 (batch 1, DIHIP_ATTN_WIDE=1: attn_plan): one workgroup
// per CU that covers 256 tokens per pass -- half the split records for the merge to wait for, the whole K / V share of a 2048-token history
// still requested up front.  Measured equal to slightly slower than the 4-wave form (profiles/r06_attn_block_polls.txt): kept as an A/B form.
constexpr int ft_mfma_smem_bytes(int nw) {
  const int epi = (nw * MF_HC * ATTN_PSTRIDE + 4) * 4, vt = nw * MF_TOK * MF_VPITCH;
  return epi > vt ? epi : vt;
}
constexpr int FT_MFMA_SMEM_BYTES = ft_mfma_smem_bytes(4);

__device__ __forceinline__ u32x4_t ld_qkv16(const uint16_t* p) { return *reinterpret_cast<const u32x4_t*>(p); }  // 16 bytes of the fused qkv row

// per-wave wall-clock stamps (tools/attn_bench on the `make trace` build); absent from the product build
#if defined(DIHIP_GEMV_TRACE) && DIHIP_GEMV_TRACE
#define DIHIP_ATTN_STAMP(I)                                                                                          \
  do {                                                                                                               \
    if (a.trace && threadIdx.x < 192) a.trace[(((size_t)bz * gy + by) * gx + bx) * 32 + (threadIdx.x >> 6) * 8 + (I)] = wall_clock64(); \
  } while (0)
// wave 0's finer stamps inside the tile loop (slots 24 .. 31: wave 3 does not stamp)
#define DIHIP_ATTN_STAMPX(I)                                                                                         \
  do {                                                                                                               \
    if (a.trace && threadIdx.x < 64) a.trace[(((size_t)bz * gy + by) * gx + bx) * 32 + 24 + (I)] = wall_clock64();   \
  } while (0)
#else
#define DIHIP_ATTN_STAMP(I) do { } while (0)
#define DIHIP_ATTN_STAMPX(I) do { } while (0)
#endif

// GATHER (fused attention block, FUSED form only): this step's q / k / v elements arrive as granules from the qkv GEMV's
// workgroups of the SAME launch (AttnHandoff): swept into an LDS image behind the kernel's buffer while the K / V tiles are in
// flight, and the merged output leaves as granules.  One head chunk per group (nchunks == 1).
constexpr int FT_MFMA_GATHER_IMG_BYTES = (MF_HC + 2) * 128 * 2;
template <int FT, int MODE, bool FUSED, bool GATHER = false, int NW = 4>
__device__ __forceinline__ void span_attn_ft_mfma_body(const AttnArgs& a, const int bx, const int by, const int bz, const int gx,
                                                       const int gy, const int gz, unsigned char* smem,
                                                       const AttnHandoff* ho = nullptr) {
  constexpr int H = 128;
  constexpr int HC = MF_HC;
  constexpr bool Q8 = MODE == DIHIP_KV_I8;
  static_assert(!FUSED || MODE == DIHIP_KV_NONE || MODE == DIHIP_KV_I8, "the decode-step form covers the 16-bit and the int8 cache");
  static_assert(!GATHER || MODE == DIHIP_KV_NONE, "the gathering form is the decode step over the 16-bit cache");
  static_assert(NW == 4 || (NW == 8 && MODE == DIHIP_KV_NONE), "the 8-wave form covers the 16-bit cache");
  constexpr int NT = NW * 64;
  constexpr int SMEM = (ft_mfma_smem_bytes(NW) + 15) & ~15;
  constexpr int ROWB = Q8 ? H : H * 2;  // bytes per token-head row in the span
  // the per-wave V tiles and the epilogue records share one buffer (a barrier separates the two uses): FT_MFMA_SMEM_BYTES
  float* lds = reinterpret_cast<float*>(smem);
  unsigned* flag_lds = reinterpret_cast<unsigned*>(lds + NW * HC * ATTN_PSTRIDE);

  const int tid = threadIdx.x, lane = tid & 63;
  DIHIP_ATTN_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kb = lane >> 4, ni = lane & 15;
  const int split = bx;
  const int grp = a.nchunks == 1 ? by : by / a.nchunks, hc = a.nchunks == 1 ? 0 : by % a.nchunks;
  const int b = bz;
  const int h0 = grp * a.hpg + hc * HC;
  const int nh = min(HC, a.hpg - hc * HC);
  unsigned char* vt = smem + wave * (MF_TOK * MF_VPITCH);
  const int lgS = 31 - __builtin_clz((unsigned)a.S);  // span lengths are powers of two (16 .. 128): shifts, not the ~25-instruction scalar division
  const void* const* ksp = a.kspans + (size_t)b * a.span_stride;
  const void* const* vsp = a.vspans + (size_t)b * a.span_stride;
  // decode-step form with a host-fixed split width: the span-table entries of this wave's first tile pair depend on the
  // kernel arguments alone -- requested here, beside the length (one round trip instead of two before the K / V loads)
  const bool spec = FUSED && a.tps_static > 0;
  int spu[2] = {0, 0};
  const void* kpu[2] = {nullptr, nullptr};
  const void* vpu[2] = {nullptr, nullptr};
  if constexpr (FUSED) {
    if (spec) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        spu[c] = min((split * a.tps_static + wave * MF_TOK + c * 16) >> lgS, a.span_stride - 1);
        kpu[c] = ksp[spu[c]];
        vpu[c] = vsp[spu[c]];
      }
    }
  }
  // (a request longer than its span table cannot be served: the length is clamped, the new token's store below is skipped)
  const int len = min((int)a.seq_lens[b] + (FUSED ? 1 : a.len_bias), a.span_stride * a.S);
  const int newpos = (int)a.seq_lens[b] + (FUSED ? 1 : a.len_bias) - 1;  // FUSED: position of this step's token
  // FUSED: the span pointers of this step's token, requested HERE: the workgroup that appends the token needs them microseconds later, and a
  // pointer load issued there is a dependent round trip (~0.45 us, timeline by split: profiles/r06_attn_block_polls.txt) in front of its tile
  // loop -- on the path of the one split every merge of the group waits for
  const void* knp = nullptr;
  const void* vnp = nullptr;
  if constexpr (FUSED && !Q8) {
    const int nsp = min(newpos >> lgS, a.span_stride - 1);
    knp = ksp[nsp];
    vnp = vsp[nsp];
  }
  const int tps = spec ? a.tps_static : (((len + a.nsplits - 1) / a.nsplits + 31) & ~31);
  const int t0 = split * tps;
  const int t1 = min(len, t0 + tps);
  const size_t par_off = (size_t)a.g * a.S * ROWB;  // int8: (zero, scale) pairs follow the data of all groups

  // K: tile c.  FT rows: k-step ks = dims ks*32 + kb*8.. (4 x 16 B per token);  int8 rows: dims kb*32.. (2 x 16 B)
  u32x4_t kreg[2][Q8 ? 2 : 4];
  // V: FT rows: load i = tokens i*4 + (lane>>4), bytes (lane&15)*16..;  int8: tokens i*8 + (lane>>3), bytes (lane&7)*16..
  u32x4_t vreg[Q8 ? 4 : 8];
  f32x4_t kpar[Q8 ? 2 : 1][2], vpar[Q8 ? 2 : 1][2];  // int8: {zero, scale} of tokens c*16 + kb*4 + {0,1 | 2,3}
  // tile bases are wave-uniform (scalar span pointer loads); tokens past the range re-read the last valid
  // token of the tile (finite data: an uninitialised row could hold NaN bit patterns, and 0 * NaN = NaN)
  auto tile_base = [&](int tb, int c, int& row0, int& last) {
    int base = tb + c * 16;
    base = base < t1 ? base : ((t1 - 1) & ~15);
    const int sp = __builtin_amdgcn_readfirstlane(base >> lgS);
    row0 = grp * a.S + (base - (sp << lgS));
    last = min(15, t1 - 1 - base);  // last valid token of the tile, relative
    return sp;
  };
  auto load_k = [&](int tb, bool first = false) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int row0, last;
      const int sp = tile_base(tb, c, row0, last);
      const unsigned char* kbase = reinterpret_cast<const unsigned char*>(first && spec && sp == spu[c] ? kpu[c] : ksp[sp]);
      const unsigned char* kd = kbase + (size_t)row0 * ROWB;
      if constexpr (Q8) {
        const uint32_t off = (uint32_t)(min(ni, last) * ROWB + kb * 32);
        kreg[c][0] = gload<u32x4_t>(kd + off);
        kreg[c][1] = gload<u32x4_t>(kd + off + 16);
        const unsigned char* kq = kbase + par_off + (size_t)row0 * 8 + kb * 32;
        kpar[c][0] = gload<f32x4_t>(kq);
        kpar[c][1] = gload<f32x4_t>(kq + 16);
      } else {
        const uint32_t off = (uint32_t)(min(ni, last) * ROWB + kb * 16);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kreg[c][ks] = gload<u32x4_t>(kd + off + ks * 64);
      }
    }
  };
  auto load_v = [&](int tb, bool first = false) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int row0, last;
      const int sp = tile_base(tb, c, row0, last);
      const unsigned char* vbase = reinterpret_cast<const unsigned char*>(first && spec && sp == spu[c] ? vpu[c] : vsp[sp]);
      const unsigned char* vd = vbase + (size_t)row0 * ROWB;
      if constexpr (Q8) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t off = (uint32_t)(min(i * 8 + (lane >> 3), last) * ROWB + (lane & 7) * 16);
          vreg[c * 2 + i] = gload<u32x4_t>(vd + off);
        }
        const unsigned char* vq = vbase + par_off + (size_t)row0 * 8 + kb * 32;
        vpar[c][0] = gload<f32x4_t>(vq);
        vpar[c][1] = gload<f32x4_t>(vq + 16);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t off = (uint32_t)(min(i * 4 + (lane >> 4), last) * ROWB + (lane & 15) * 16);
          vreg[c * 4 + i] = gload<u32x4_t>(vd + off);
        }
      }
    }
  };
  // 4 int8 bytes of a dword -> 4 exact FT values 128 + q (two packed dwords)
  auto expand8 = [&](uint32_t d0, uint32_t d1) {  // 8 bytes -> one MFMA operand register quad
    const uint32_t u0 = d0 ^ 0x80808080u, u1 = d1 ^ 0x80808080u;
    return u32x4_t{pack_ft2<FT>((float)(u0 & 0xFFu), (float)((u0 >> 8) & 0xFFu)),
                   pack_ft2<FT>((float)((u0 >> 16) & 0xFFu), (float)(u0 >> 24)),
                   pack_ft2<FT>((float)(u1 & 0xFFu), (float)((u1 >> 8) & 0xFFu)),
                   pack_ft2<FT>((float)((u1 >> 16) & 0xFFu), (float)(u1 >> 24))};
  };

  const int tb0 = t0 + wave * MF_TOK;
  const bool active = tb0 < t1;
  if (active) {
    load_v(tb0, true);
    load_k(tb0, true);
  }
  DIHIP_ATTN_STAMP(1);

  // rotate-half on a lane's fragments: dims ks*32 + kb*8 + e (ks = 0, 1) pair with ks + 2; table row = position.
  // Same arithmetic and rounding as dihip_rope_qk / the Rotary op: two products, one add, rounded to FT.
  // (GATHER: the sweep of the granules rotates -- see there)
  auto rotate = [&](u32x4_t (&f)[4], const float* cs_row) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f32x4_t* t = reinterpret_cast<const f32x4_t*>(cs_row + (ks * 32 + kb * 8) * 2);
      const f32x4_t c0 = t[0], c1 = t[1], c2 = t[2], c3 = t[3];  // {cos, sin} x 8 dims
      const float cosv[8] = {c0[0], c0[2], c1[0], c1[2], c2[0], c2[2], c3[0], c3[2]};
      const float sinv[8] = {c0[1], c0[3], c1[1], c1[3], c2[1], c2[3], c3[1], c3[3]};
      u32x4_t lo_, hi_;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x0[2], x1[2], r0[2], r1[2];
        x0[0] = ft_bits_to_f32<FT>(f[ks][j] & 0xFFFFu);
        x0[1] = ft_bits_to_f32<FT>(f[ks][j] >> 16);
        x1[0] = ft_bits_to_f32<FT>(f[ks + 2][j] & 0xFFFFu);
        x1[1] = ft_bits_to_f32<FT>(f[ks + 2][j] >> 16);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          r0[e] = x0[e] * cosv[2 * j + e] - x1[e] * sinv[2 * j + e];
          r1[e] = x1[e] * cosv[2 * j + e] + x0[e] * sinv[2 * j + e];
        }
        lo_[j] = f32_to_ft_bits<FT>(r0[0]) | (f32_to_ft_bits<FT>(r0[1]) << 16);
        hi_[j] = f32_to_ft_bits<FT>(r1[0]) | (f32_to_ft_bits<FT>(r1[1]) << 16);
      }
      f[ks] = lo_;
      f[ks + 2] = hi_;
    }
  };
  // Q as the B operand, unscaled (exact FT values): lane (kb, head ni) holds the 8 dims of k-step ks in the order
  // of the K fragments (FT rows: ks*32 + kb*8..; int8 rows: kb*32 + ks*8..)
  u32x4_t qf[4];
  float qsum = 0.f;
  const size_t qrow_stride = FUSED ? (size_t)(a.n + 2 * a.g) * H : (size_t)a.n * H;
  const float* cs_row = FUSED ? a.rope_tab + (size_t)newpos * 128 : nullptr;
  uint16_t* const img = reinterpret_cast<uint16_t*>(smem + SMEM);  // GATHER: [hpg q heads][k][v] x 128
  if constexpr (GATHER) {
    static_assert(FUSED && MODE == DIHIP_KV_NONE, "the gathering form is the decode step over the 16-bit cache");
    // The sweep ROTATES: a thread takes the two granules of a rotate-half pair (dims d, d + 64 of a query head or of this step's K head),
    // applies the Rotary arithmetic of rotate() below -- two products, one add, rounded to FT -- once, and the LDS image holds rotated
    // rows.  (Round 5 / 6 rotated the fragments: ~256 instructions in EVERY wave after the hand-off, and as many again for the K head in
    // the workgroup of the last split -- the one every merge waits for.)  The {cos, sin} pair depends on d = tid % 64 alone: requested
    // before the wait.  V granules (no rotation): one more load for 128 threads.
    // The V tile of the first pass goes to LDS BEFORE the wait (it does not depend on q; the new token's rows are patched in the loop)
#ifndef DIHIP_AB_VEARLY
#define DIHIP_AB_VEARLY 1
#endif
    if constexpr (DIHIP_AB_VEARLY) {
      if (active) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          *reinterpret_cast<u32x4_t*>(smem + wave * (MF_TOK * MF_VPITCH) + (i * 4 + (lane >> 4)) * MF_VPITCH + (lane & 15) * 16) = vreg[i];
      }
    }
    const int npair = (a.hpg + 1) * 64;
    constexpr int GP = NW == 8 ? 1 : 2;  // pairs per thread and round: (7 + 1) heads x 64 = 1 x 512 threads = 2 x 256
    const float cs[2] = {cs_row[(tid & 63) * 2], cs_row[(tid & 63) * 2 + 1]};
    for (int base = 0; base < npair; base += GP * NT) {
      int s0[GP], s1[GP];
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const int pi = min(base + j * NT + tid, npair - 1), hh = pi >> 6, d = pi & 63;
        s0[j] = (hh < a.hpg ? (h0 + hh) * H : (a.n + grp) * H) + d;
        s1[j] = s0[j] + 64;
      }
      const bool has_v = base == 0 && tid < H;
      const int sv = (a.n + a.g + grp) * H + (has_v ? tid : 0);
      unsigned long long g0[GP], g1[GP], gvv;
      for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < GP; ++j) {
          g0[j] = __hip_atomic_load(ho->qkv_gran + s0[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          g1[j] = __hip_atomic_load(ho->qkv_gran + s1[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && (unsigned)(g0[j] >> 32) == ho->tag && (unsigned)(g1[j] >> 32) == ho->tag;
        }
        gvv = __hip_atomic_load(ho->qkv_gran + sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && (unsigned)(gvv >> 32) == ho->tag;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        if (spins > ho->spin_limit) {  // gives up (wave-uniform): garbage results, flagged, never a hang
          if (lane == 0) __hip_atomic_store(ho->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
#ifndef DIHIP_AB_SLEEP_Q
#define DIHIP_AB_SLEEP_Q 1
#endif
        __builtin_amdgcn_s_sleep(DIHIP_AB_SLEEP_Q);
      }
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const int pi = base + j * NT + tid;
        if (pi < npair) {
          const float x0 = ft_bits_to_f32<FT>((uint32_t)g0[j] & 0xFFFFu), x1 = ft_bits_to_f32<FT>((uint32_t)g1[j] & 0xFFFFu);
          const float r0 = x0 * cs[0] - x1 * cs[1];
          const float r1 = x1 * cs[0] + x0 * cs[1];
          const int at = (pi >> 6) * H + (pi & 63);
          img[at] = (uint16_t)f32_to_ft_bits<FT>(r0);
          img[at + 64] = (uint16_t)f32_to_ft_bits<FT>(r1);
        }
      }
      if (has_v) img[(a.hpg + 1) * H + tid] = (uint16_t)gvv;
    }
    __syncthreads();
    DIHIP_ATTN_STAMPX(4);  // granules swept into the LDS image
  }
  {
    const bool hv = ni < nh;
    const uint16_t* qrow = GATHER ? img + (size_t)(hv ? ni : 0) * H
                                  : reinterpret_cast<const uint16_t*>(a.q) + (size_t)b * qrow_stride + (size_t)(h0 + (hv ? ni : 0)) * H;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = ld_qkv16(qrow + (Q8 ? kb * 32 + ks * 8 : ks * 32 + kb * 8));
      if constexpr (Q8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) qsum += hv ? ft_bits_to_f32<FT>(qf[ks][j] & 0xFFFFu) + ft_bits_to_f32<FT>(qf[ks][j] >> 16) : 0.f;
      }
    }
    if constexpr (FUSED && Q8) {
      // int8 rows order the dims kb*32 + ks*8 + e: the rotate-half partner d +- 64 sits in lane +- 32 (same ks, same dword) --
      // one cross-half exchange per dword; products, sum and FT rounding as dihip_rope_qk / the Rotary op
      qsum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        u32x4_t rot;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t pw = (uint32_t)__shfl_xor((int)qf[ks][j], 32, 64);
          const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(cs_row + (size_t)((kb & 1) * 32 + ks * 8 + 2 * j) * 2);  // {c0, s0, c1, s1}
          float r[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float x = ft_bits_to_f32<FT>(e ? qf[ks][j] >> 16 : qf[ks][j] & 0xFFFFu);
            const float pt = ft_bits_to_f32<FT>(e ? pw >> 16 : pw & 0xFFFFu);
            r[e] = kb < 2 ? x * cs[2 * e] - pt * cs[2 * e + 1] : x * cs[2 * e] + pt * cs[2 * e + 1];
          }
          rot[j] = f32_to_ft_bits<FT>(r[0]) | (f32_to_ft_bits<FT>(r[1]) << 16);
          qsum += hv ? ft_bits_to_f32<FT>(rot[j] & 0xFFFFu) + ft_bits_to_f32<FT>(rot[j] >> 16) : 0.f;  // (of the ROTATED row, as the two-launch form sums it)
        }
        qf[ks] = rot;
      }
    } else if constexpr (FUSED && !GATHER) {
      rotate(qf, cs_row);  // (GATHER: the image holds rotated rows)
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      if (!hv) qf[ks] = u32x4_t{0u, 0u, 0u, 0u};
    if constexpr (Q8) {
      qsum = rows_sum(qsum);
    }
  }
  // FUSED: this step's K (rotated, rounded) and V head of the group, as the cache will hold them
  const bool has_new = FUSED && newpos >= t0 && newpos < t0 + tps;  // workgroup-uniform (all head chunks of the group)
  u32x4_t knew[4] = {}, vnew = {};
  // int8: the new K / V head rotated, rounded and QUANTISED as the append kernel does (store_token_head) -- every wave for itself into
  // its own 2 x 144 bytes of LDS ({128 codes, zero, scale}; no workgroup barrier: a wave reads back what it wrote itself, LDS
  // operations of a wave execute in order); the tile loop substitutes codes and parameters where a lane's token is the new one
  unsigned char* const nrow = smem + SMEM + wave * 288;
  if constexpr (FUSED && Q8) {
    if (has_new) {
      const int sp = newpos >> lgS, pos = newpos - (sp << lgS);
      const bool writer = hc == 0 && wave == 0 && sp < a.span_stride;  // one per (request, group): DecoderCacheAppend; past the span table: dropped
#pragma unroll
      for (int kv = 0; kv < 2; ++kv) {
        const uint16_t* row = reinterpret_cast<const uint16_t*>(a.q) + (size_t)b * qrow_stride + (size_t)(a.n + (kv ? a.g : 0) + grp) * H;
        const uint32_t w = reinterpret_cast<const uint32_t*>(row)[lane];  // lane holds d = 2 * lane, 2 * lane + 1 (the codec's element order)
        float x[2] = {ft_bits_to_f32<FT>(w & 0xFFFFu), ft_bits_to_f32<FT>(w >> 16)};
        if (kv == 0) {
          const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(cs_row + (size_t)((2 * lane) & 63) * 2);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float pt = __shfl_xor(x[e], 32, 64);
            const float r = lane < 32 ? x[e] * cs[2 * e] - pt * cs[2 * e + 1] : x[e] * cs[2 * e] + pt * cs[2 * e + 1];
            x[e] = ft_round<FT>(r);
          }
        }
        unsigned char* nr = nrow + kv * 144;
        store_token_head<FT, DIHIP_KV_I8, 2>(nr, x, 0, 0, 1, 1, H, lane);  // a one-token "span": 128 codes, then {zero, scale}
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (writer) {  // the same bytes into the span
          unsigned char* span = reinterpret_cast<unsigned char*>(const_cast<void*>((kv ? vsp : ksp)[sp]));
          const size_t rowi = (size_t)grp * a.S + pos;
          if (lane < 32) gstore<uint32_t>(span + rowi * ROWB + lane * 4, reinterpret_cast<const uint32_t*>(nr)[lane]);
          if (lane == 32) gstore<uint64_t>(span + par_off + rowi * 8, *reinterpret_cast<const uint64_t*>(nr + H));
        }
      }
    }
  }
  if constexpr (FUSED && !Q8) {
    if (has_new) {
      const uint16_t* krow = GATHER ? img + (size_t)a.hpg * H
                                    : reinterpret_cast<const uint16_t*>(a.q) + (size_t)b * qrow_stride + (size_t)(a.n + grp) * H;
      const uint16_t* vrow = GATHER ? krow + H : krow + (size_t)a.g * H;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) knew[ks] = ld_qkv16(krow + ks * 32 + kb * 8);
      if constexpr (!GATHER) rotate(knew, cs_row);
      vnew = ld_qkv16(vrow + (lane & 15) * 8);
      if (hc == 0 && wave == 0 && (newpos >> lgS) < a.span_stride) {  // one writer per (request, group): DecoderCacheAppend; a token
        // past the span table is dropped, as kv_append_kernel does (span_cache.hip) -- never written over a cached one
        const int sp = newpos >> lgS, pos = newpos - (sp << lgS);
        unsigned char* kd = reinterpret_cast<unsigned char*>(const_cast<void*>(knp)) + ((size_t)grp * a.S + pos) * ROWB;
        unsigned char* vd = reinterpret_cast<unsigned char*>(const_cast<void*>(vnp)) + ((size_t)grp * a.S + pos) * ROWB;
        if (ni == 0) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) gstore<u32x4_t>(kd + ks * 64 + kb * 16, knew[ks]);
        }
        if (lane < 16) gstore<u32x4_t>(vd + lane * 16, vnew);
      }
    }
  }

  DIHIP_ATTN_STAMP(2);
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f, czero = 0.f;
  f32x4_t o[8];  // O^T tile dt: rows (dims) dt*16 + kb*4 + r, column = head ni
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) o[dt] = zero4;
  // transpose-read addresses: lane p of a 16-lane group supplies row p/4 (token), columns (p%4)*4.. of a 4 x 16 block
  const unsigned char* tr0 = vt + (kb * 4 + (ni >> 2)) * MF_VPITCH + (ni & 3) * 8;

#ifdef DIHIP_AB_VEARLY
  constexpr bool v_staged = GATHER && DIHIP_AB_VEARLY;  // the first pass's V tile was stored before the wait for q
#else
  constexpr bool v_staged = false;
#endif
  if (active) {
    constexpr int STEP = NW * MF_TOK;
    for (int tb = tb0; tb < t1; tb += STEP) {
      // FUSED: lanes whose (clamped) token is this step's token take the register copy (see above)
      if constexpr (FUSED && Q8) {
        if (has_new) {
          const float knz = reinterpret_cast<const float*>(nrow + H)[0], kns = reinterpret_cast<const float*>(nrow + H)[1];
          const float vnz = reinterpret_cast<const float*>(nrow + 144 + H)[0], vns = reinterpret_cast<const float*>(nrow + 144 + H)[1];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            int base = tb + c * 16;
            base = base < t1 ? base : ((t1 - 1) & ~15);
            const int last = min(15, t1 - 1 - base);
            if (base + min(ni, last) == newpos) {  // K codes: this lane's 32 dims of the row
              kreg[c][0] = *reinterpret_cast<const u32x4_t*>(nrow + kb * 32);
              kreg[c][1] = *reinterpret_cast<const u32x4_t*>(nrow + kb * 32 + 16);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
              if (base + kb * 4 + rr == newpos) {  // parameters of token c*16 + kb*4 + rr (loaded unclamped)
                kpar[c][rr >> 1][(rr & 1) * 2] = knz;
                kpar[c][rr >> 1][(rr & 1) * 2 + 1] = kns;
                vpar[c][rr >> 1][(rr & 1) * 2] = vnz;
                vpar[c][rr >> 1][(rr & 1) * 2 + 1] = vns;
              }
#pragma unroll
            for (int i = 0; i < 2; ++i)
              if (base + min(i * 8 + (lane >> 3), last) == newpos) vreg[c * 2 + i] = *reinterpret_cast<const u32x4_t*>(nrow + 144 + (lane & 7) * 16);
          }
        }
      }
      if constexpr (FUSED && !Q8) {
        if (has_new) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            int base = tb + c * 16;
            base = base < t1 ? base : ((t1 - 1) & ~15);
            const int last = min(15, t1 - 1 - base);
            if (base + min(ni, last) == newpos) {
#pragma unroll
              for (int ks = 0; ks < 4; ++ks) kreg[c][ks] = knew[ks];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (base + min(i * 4 + (lane >> 4), last) == newpos) {
                if (v_staged && tb == tb0)  // (the tile sits in LDS already: patch the row)
                  *reinterpret_cast<u32x4_t*>(vt + ((c * 4 + i) * 4 + (lane >> 4)) * MF_VPITCH + (lane & 15) * 16) = vnew;
                else
                  vreg[c * 4 + i] = vnew;
              }
          }
        }
      }
      // ---- V(t) into this wave's LDS tile (FT elements, [token][dim]), then request V(t + 1) into the same registers
      float vzp[2][4], vsc[2][4];  // int8: parameters of this lane's tokens (c, kb*4 + rr), read before the refill
      if constexpr (Q8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned char* dst = vt + (i * 8 + (lane >> 3)) * MF_VPITCH + (lane & 7) * 32;
          *reinterpret_cast<u32x4_t*>(dst) = expand8(vreg[i][0], vreg[i][1]);
          *reinterpret_cast<u32x4_t*>(dst + 16) = expand8(vreg[i][2], vreg[i][3]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            vzp[c][rr] = vpar[c][rr >> 1][(rr & 1) * 2];
            vsc[c][rr] = vpar[c][rr >> 1][(rr & 1) * 2 + 1];
          }
      } else {
        if (!(v_staged && tb == tb0)) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            *reinterpret_cast<u32x4_t*>(vt + (i * 4 + (lane >> 4)) * MF_VPITCH + (lane & 15) * 16) = vreg[i];
        }
      }
      DIHIP_ATTN_STAMPX(0);  // V tile written to LDS
      load_v(tb + STEP);
      // ---- scores of the 32 tokens (transposed): sc[c][r] = token tb + c*16 + kb*4 + r, head ni
      float sc[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x4_t acc = zero4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if constexpr (Q8) {
            acc = mfma_ft<FT>(expand8(kreg[c][ks >> 1][(ks & 1) * 2], kreg[c][ks >> 1][(ks & 1) * 2 + 1]), qf[ks], acc);
          } else {
            acc = mfma_ft<FT>(kreg[c][ks], qf[ks], acc);
          }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float v = acc[rr] * a.scale;
          if constexpr (Q8) {
            const float kz = kpar[c][rr >> 1][(rr & 1) * 2], ksc = kpar[c][rr >> 1][(rr & 1) * 2 + 1];
            v = (ksc * a.scale) * fmaf(-(128.f + kz), qsum, acc[rr]);
          }
          sc[c][rr] = tb + c * 16 + kb * 4 + rr < t1 ? v : -INFINITY;
        }
      }
      DIHIP_ATTN_STAMPX(1);  // scores done
      load_k(tb + STEP);
      // ---- online softmax (lane-local + two cross-row shuffles)
      float mn = m;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) mn = fmaxf(mn, sc[c][rr]);
      mn = rows_max(mn);
      const float corr = safe_exp_diff(m, mn);
      const float m_old = m;
      m = mn;
      float ps = 0.f, cz = 0.f;
      uint32_t pk[4], pl[4];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          float pv[2], zz[2] = {0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int rr = 2 * h2 + e;
            const float p_ = safe_exp_diff(sc[c][rr], mn);
            ps += p_;
            pv[e] = p_;
            if constexpr (Q8) {
              const bool valid = tb + c * 16 + kb * 4 + rr < t1;  // p == 0 there, but the parameters may be junk
              pv[e] = valid ? p_ * vsc[c][rr] : 0.f;
              zz[e] = valid ? 128.f + vzp[c][rr] : 0.f;
            }
          }
          const uint32_t hi = pack_ft2<FT>(pv[0], pv[1]);
          const float h0f = ft_bits_to_f32<FT>(hi & 0xFFFFu), h1f = ft_bits_to_f32<FT>(hi >> 16);
          const uint32_t lo = pack_ft2<FT>(pv[0] - h0f, pv[1] - h1f);
          pk[c * 2 + h2] = hi;
          pl[c * 2 + h2] = lo;
          if constexpr (Q8) {
            cz = fmaf(h0f + ft_bits_to_f32<FT>(lo & 0xFFFFu), zz[0], cz);
            cz = fmaf(h1f + ft_bits_to_f32<FT>(lo >> 16), zz[1], cz);
          }
        }
      l = l * corr + ps;
      czero = czero * corr + cz;
      // (a first tile -- m_old = -inf in every lane, o still all zero -- skips the 32 multiplications by corr = 0: same bits, o stays +0)
      if (__builtin_amdgcn_ballot_w64(corr != 1.f && m_old != -INFINITY) != 0ull) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) o[dt][rr] *= corr;
      }
      DIHIP_ATTN_STAMPX(2);  // softmax done
      // ---- O^T += V^T . P: A = V^T from the LDS tile by transpose reads; k-slot j = token (j>>2)*16 + kb*4 + (j&3)
      const u32x4_t pkv = {pk[0], pk[1], pk[2], pk[3]};
      const u32x4_t plv = {pl[0], pl[1], pl[2], pl[3]};
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const u32x2_t lo2 = lds_read_tr16(tr0 + dt * 32);
        const u32x2_t hi2 = lds_read_tr16(tr0 + dt * 32 + 16 * MF_VPITCH);
        const u32x4_t vf = {lo2[0], lo2[1], hi2[0], hi2[1]};
        o[dt] = mfma_ft<FT>(vf, pkv, o[dt]);
        o[dt] = mfma_ft<FT>(vf, plv, o[dt]);
      }
    }
  }
  DIHIP_ATTN_STAMPX(3);  // loop left
  l = rows_sum(l);
  if constexpr (Q8) {
    czero = rows_sum(czero);
  }
  DIHIP_ATTN_STAMP(3);
  __syncthreads();  // every wave is done with its V tile: the buffer now holds the epilogue records
  if (ni < nh) {
    float* rec = lds + (wave * HC + ni) * ATTN_PSTRIDE;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      f32x4_t x = o[dt];
      if constexpr (Q8) x = f32x4_t{x[0] - czero, x[1] - czero, x[2] - czero, x[3] - czero};
      *reinterpret_cast<f32x4_t*>(rec + dt * 16 + kb * 4) = x;
    }
    if (kb == 0) {
      rec[H] = m;
      rec[H + 1] = l;
    }
  }
  if constexpr (GATHER) {
    // the fused block: polled split records (nsplits <= 32, the host's contract), no drain, no ticket.  (The ticket protocol below is not
    // compiled into the block any more: its 32-wide reload kept the whole kernel at the register ceiling.)
    attn_block_epilogue_polled<FT, HC, NW>(a, lds, b, h0, nh, split,
                                           a.trace && threadIdx.x < 192 ? a.trace + (((size_t)bz * gy + by) * gx + bx) * 32 + (threadIdx.x >> 6) * 8 : nullptr, ho);
    DIHIP_ATTN_STAMP(7);
    return;
  } else if constexpr (FUSED) {
    if (a.merge_wt) {
      attn_block_epilogue_wt<FT, HC, GATHER, NW>(a, lds, flag_lds, b, h0, nh, split,
                                                 a.counters + (((size_t)b * a.g + grp) * a.nchunks + hc) * 32,  // one 128-byte line each
                                                 a.trace && threadIdx.x < 192 ? a.trace + (((size_t)bz * gy + by) * gx + bx) * 32 + (threadIdx.x >> 6) * 8 : nullptr,
                                                 ho, grp);
      DIHIP_ATTN_STAMP(7);
      return;
    }
  }
  attn_block_epilogue<FT, HC, NW>(a, lds, flag_lds, b, h0, nh, split);
  DIHIP_ATTN_STAMP(7);
}


template <int FT, int MODE, bool FUSED>
__global__ __launch_bounds__(ATTN_THREADS, 2) void span_attn_ft_mfma_kernel(const AttnArgs a) {
  // (+ the int8 decode step's per-wave new-row buffers: 4 x 2 x 144 bytes behind the tiles)
  __shared__ __attribute__((aligned(16))) unsigned char smem[FT_MFMA_SMEM_BYTES + (FUSED && MODE == DIHIP_KV_I8 ? 16 + 4 * 288 : 0)];
  span_attn_ft_mfma_body<FT, MODE, FUSED>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z, smem);
}

// the 8-wave form (16-bit cache, batch 1: AttnPlan::waves): 72 KB of V tiles -- dynamic LDS, ft_mfma_smem_bytes(8)
template <int FT, bool FUSED>
__global__ __launch_bounds__(512) void span_attn_ft_mfma_w8_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  span_attn_ft_mfma_body<FT, DIHIP_KV_NONE, FUSED, false, 8>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z, dyn_smem);
}

}  // namespace dihip
