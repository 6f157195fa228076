// prefill_attn.hip -- causal GQA prefill attention on the gfx950 matrix cores
// (include/dashinfer_hip.h section 4).
//
// Replaces xformer_prefill_attention (csrc/core/kernel/cuda/xformer_mha/xformer_mha.h:26-41; CUTLASS
// fMHA with kBlockQuery = 32, kBlockKey = 128, xformer_mha.dispatch.cu:37-41) with a
// flash-style single pass written for wave64 MFMA:
//   * workgroup = 4 waves = 128 query rows of one head; each wave owns 32 rows (two 16-row MFMA
//     tiles sharing every K / V fragment read) and keeps its Q fragments and the 32 x 128 f32 output
//     accumulator in registers for the whole pass.  One-dimensional grid, longest query tiles first: in-order
//     dispatch then balances the causal triangle greedily;
//   * K and V tiles of 64 keys are staged once per workgroup in LDS, both row-major as they come (buffer loads with the
//     hardware range check -> registers -> 16-byte LDS stores; two copies, so one barrier per tile; the loads of tile
//     T + 2 leave right after the first MFMA phase of tile T).  K rows are the A fragments of K.Q^T as stored, 16-byte
//     chunks XOR-swizzled by the row (conflict-free ds_read_b128); the A fragments of V^T.P^T are read with the
//     transposing LDS load (ds_read_b64_tr_b16) in the k-slot order P leaves the first MFMA;
//   * TRANSPOSED formulation, S^T = K.Q^T and O^T = V^T.P^T: a lane owns one query per 16-query tile and
//     its accumulator rows are keys / head dims.  The online softmax is lane-local (f32: causal mask on diagonal tiles
//     only -> max over the lane's 16 scores + two v_permlane swaps -> exp2 with alpha * log2(e) folded into one fma ->
//     P rounded to FT by the packed hardware convert), P leaves the S^T accumulators already in the B-operand layout
//     of the second MFMA (no LDS round trip), the running rescale is skipped while no maximum moves, and the
//     output is 4 consecutive dims per lane (8-byte stores);
//   * per 64 MFMAs a wave issues ~180 VALU instructions (2.8 per MFMA; the first version had 4.4 on top of a VALU
//     transpose in the staging) -- the VALU issue port, not the matrix pipe, is what bounds this kernel;
//   * key tiles entirely above the causal diagonal are skipped.
// Numerics follow the reference's CPU check (tests/cpp/kernel/cuda/kernel_mhaprefill_test.cpp:
// 119-320): f32 softmax, FT inputs/outputs; P is rounded to FT before P.V as the tensor-core
// kernels of the reference do.
#include <algorithm>
#include <cstdlib>

#include "gemm_lowp_kernel.hpp"  // mfma16<FT>

namespace dihip {

constexpr int PF_THREADS = 256;
// MT = 16-row query tiles per wave (template): 2 -> 128 query rows per workgroup, every K / V fragment read from
// LDS feeds two MFMAs; 1 -> 64 rows per workgroup, used while the larger tile would leave CUs without work
constexpr int PF_KEYS = 64;    // keys per tile (4 MFMA key sub-tiles: the per-tile softmax bookkeeping is paid per 64 keys)
constexpr int PF_KPITCH = 128; // K tile row pitch in elements; the 16-byte chunk c of row r sits at chunk c ^ (r & 15):
                               // ds_read_b128 serves lanes in groups {0-3,12-15,20-27}.. (two k-chunks x 8 rows each), which
                               // row padding cannot spread over the 64 banks but the XOR does (0 conflicts)
constexpr int PF_VPITCH = 144; // V tile row pitch in elements (128 + 16: conflict-free transposing reads)

struct PrefillArgs {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  int seq_q, seq_k, q_stride, kv_stride, n_heads, n_groups, causal;
  float alpha;
  int prio;
};

// NG = key groups per workgroup: 1 -> 4 waves, two workgroups per CU; 2 -> 8 waves, one workgroup per CU whose two
// 4-wave groups take alternate key tiles of the SAME query tile and merge (O, m, l) through LDS at the end.  The CU
// holds the same 8 waves either way, but the chain of dependent key tiles per query tile is half as long: that is what
// bounds short prompts, where every workgroup is resident from the start and the longest one sets the time.
template <int FT, int MT, int NG>
__global__ __launch_bounds__(PF_THREADS * NG, NG == 1 ? 2 : 1) void prefill_attn_kernel(const PrefillArgs a) {
  constexpr int H = 128;
  constexpr int PF_QROWS = 64 * MT;  // query rows per workgroup (4 waves x 16 x MT)
  // two copies of each tile: the next tile is written while the current one is read, one barrier per tile
  __shared__ __attribute__((aligned(16))) uint16_t ks_all[NG][2][PF_KEYS * PF_KPITCH];   // K tile [key][dim]
  __shared__ __attribute__((aligned(16))) uint16_t vs_all[NG][2][PF_KEYS * PF_VPITCH];   // V tile [key][dim]

  const int lane = threadIdx.x & 63;
  const int wave_wg = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // uniform: per-wave tile tests become scalar branches
  const int grp = NG == 1 ? 0 : wave_wg >> 2;  // key group of this wave
  const int wave = wave_wg & 3;                // its 16 * MT query rows inside the workgroup's tile
  const int tid = threadIdx.x & (PF_THREADS - 1);  // thread index inside the group (staging)
  uint16_t (*ks_buf)[PF_KEYS * PF_KPITCH] = ks_all[grp];
  uint16_t (*vs_buf)[PF_KEYS * PF_VPITCH] = vs_all[grp];
  const int ni = lane & 15, kb = lane >> 4;
  // The grid is one-dimensional, heads fastest, and under the causal mask the LONGEST query tiles come first: workgroups
  // are dispatched in index order to whichever CU frees up, i.e. greedy longest-job-first balancing of the triangle
  // (measured against pairing each query tile with its mirror: +9 % at 8192, +2..4 % elsewhere)
  const int head = (int)blockIdx.x % a.n_heads;
  const int kvh = head / (a.n_heads / a.n_groups);
  const int shift = a.seq_k - a.seq_q;  // query i sees keys j <= i + shift
  const int nqt = (a.seq_q + PF_QROWS - 1) / PF_QROWS;
  constexpr int NT = PF_KEYS / 16;  // 16-key MFMA sub-tiles per key tile
  const int row_bytes = a.kv_stride * 2;
  const int kv_bytes = (a.seq_k - 1) * row_bytes + H * 2;  // the kv head's rows, from its first element
  const int voff = (tid >> 4) * row_bytes + (tid & 15) * 16;

  {
    const int qt_lin = (int)blockIdx.x / a.n_heads;
    const int qt = a.causal ? nqt - 1 - qt_lin : qt_lin;
    const int q0 = qt * PF_QROWS + wave * (16 * MT);  // first query row of this wave

    // Q fragments: lane (kb, ni) holds Q[q0 + mt*16 + ni][s*32 + kb*8 .. +8]
    u32x4_t qf[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int qr = min(q0 + mt * 16 + ni, a.seq_q - 1);
      const uint16_t* qp = reinterpret_cast<const uint16_t*>(a.q) + (size_t)qr * a.q_stride + (size_t)head * H + kb * 8;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[mt][s] = *reinterpret_cast<const u32x4_t*>(qp + s * 32);
    }
    // Transposed formulation (S^T = K.Q^T, O^T = V^T.P^T): a lane owns ONE query (column ni) per 16-query tile,
    // its accumulator rows are keys / head dims.  The online softmax is then lane-local (8 scores per tile in
    // registers, two cross-row shuffles for the max), P leaves the S^T accumulators already in the B-operand
    // layout of the P.V MFMA (no LDS round trip), and the output is 4 consecutive dims per lane.
    f32x4_t oacc[MT][8];       // O^T tile t: rows (dims) t*16 + kb*4 + r, column = query ni
    float mrow[MT], lrow[MT];  // running max (uniform over kb) / partial sum (this lane's keys) of query ni
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int t = 0; t < 8; ++t) oacc[mt][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      mrow[mt] = -1e30f;
      lrow[mt] = 0.f;
    }
    // keys needed by this query tile: up to the diagonal of its last row
    const int wg_last_q = min(a.seq_q, (qt + 1) * PF_QROWS) - 1;
    const int k_end = a.causal ? min(a.seq_k, wg_last_q + shift + 1) : a.seq_k;
    const float sl2 = a.alpha * 1.44269504088896341f;  // exp(alpha * s) = exp2(sl2 * s)

    // global -> register prefetch of one K/V tile (T + 2 while tile T is consumed): buffer loads with a per-thread
    // byte offset fixed for the whole pass and the tile's row offset in an SGPR -- no address arithmetic in the loop --
    // and rows past seq_k read as zeros (hardware range check), so neither loads nor LDS writes need a guard.
    const auto rsrc_k = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(reinterpret_cast<const uint16_t*>(a.k) + (size_t)kvh * H), 0, kv_bytes, 0x00020000);
    const auto rsrc_v = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(reinterpret_cast<const uint16_t*>(a.v) + (size_t)kvh * H), 0, kv_bytes, 0x00020000);
    u32x4_t kreg[4], vreg[4];  // thread -> keys (tid >> 4) + 16 * it of the tile, 16-byte dim chunk tid & 15
    auto load_tile = [&](int k0) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int soff = (k0 + 16 * it) * row_bytes;
        kreg[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_k, voff, soff, 0);
        vreg[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, voff, soff, 0);
      }
    };
    auto stage_tile = [&](int buf) {
      const int r = tid >> 4, dc = tid & 15;  // row r + 16 * it: the XOR key r & 15 is the same for the four rows
      uint16_t* ksw = ks_buf[buf] + r * PF_KPITCH + (dc ^ r) * 8;
      uint16_t* vsw = vs_buf[buf] + r * PF_VPITCH + dc * 8;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        *reinterpret_cast<u32x4_t*>(ksw + it * 16 * PF_KPITCH) = kreg[it];
        *reinterpret_cast<u32x4_t*>(vsw + it * 16 * PF_VPITCH) = vreg[it];
      }
    };
    // group g walks key tiles g, g + NG, ...: tile T of the group starts at key kstep * T + koff
    constexpr int kstep = NG * PF_KEYS;
    const int koff = grp * PF_KEYS;
    load_tile(koff);
    stage_tile(0);
    load_tile(koff + kstep);
    __syncthreads();
    int buf = 0;
    for (int kbase = 0; kbase < k_end; kbase += kstep, buf ^= 1) {
      const int k0 = kbase + koff;
      // the other copy was last read one iteration ago and every wave has passed that iteration's barrier
      stage_tile(buf ^ 1);
      const uint16_t* ks = ks_buf[buf];
      // transposing reads of the V tile (ds_read_b64_tr_b16): lane p of a 16-lane group supplies row p/4 (key kb*4 + p/4),
      // columns (p%4)*4.. of a 4 x 16 block and receives column p%16 of it, i.e. 4 keys of ONE head dim
      const unsigned char* tr0 = reinterpret_cast<const unsigned char*>(vs_buf[buf]) + (kb * 4 + (ni >> 2)) * (PF_VPITCH * 2) + (ni & 3) * 8;
      // some key of the tile is visible to this wave (NG = 2: the other group's last tile may lie past k_end)
      const bool wave_live = k0 < k_end && (!a.causal || k0 <= q0 + 16 * MT - 1 + shift);
      if (wave_live && q0 < a.seq_q) {
        // ---- S^T = K.Q^T: the K fragments (A) of a key sub-tile feed all MT query tiles (B).  All 16 fragment reads are
        //      issued before the first MFMA (one LDS latency per tile) and the NT x MT accumulators are independent chains ----
        u32x4_t kf[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            kf[nt][s] = *reinterpret_cast<const u32x4_t*>(ks + (nt * 16 + ni) * PF_KPITCH + ((kb + 4 * s) ^ ni) * 8);
        __builtin_amdgcn_sched_barrier(0);
        if (a.prio) __builtin_amdgcn_s_setprio(1);  // the matrix phases win the issue arbitration against the other workgroup's softmax
        f32x4_t sacc[MT][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) sacc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) sacc[mt][nt] = mfma16<FT>(kf[nt][s], qf[mt][s], sacc[mt][nt]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // the K fragment registers are free again: tile T + 2 leaves for the prefetch registers now and has the softmax,
        // the second MFMA phase and the barrier to land
        load_tile(k0 + 2 * kstep);
        // V fragments of the first four dim tiles: the transposing reads fly while the softmax runs on the VALU
        // (which has no LDS traffic of its own).  Fragment hh covers keys hh*32 ..+32 in the k-slot order of P.
        u32x4_t vf[8][2];
        auto read_v = [&](int t) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const u32x2_t lo2 = lds_read_tr16(tr0 + t * 32 + (hh * 32) * PF_VPITCH * 2);
            const u32x2_t hi2 = lds_read_tr16(tr0 + t * 32 + (hh * 32 + 16) * PF_VPITCH * 2);
            vf[t][hh] = u32x4_t{lo2[0], lo2[1], hi2[0], hi2[1]};
          }
        };
#pragma unroll
        for (int t = 0; t < 4; ++t) read_v(t);
        __builtin_amdgcn_sched_barrier(0);
        // ---- mask + online softmax: lane holds S[q = q0 + mt*16 + ni][key = k0 + nt*16 + kb*4 + r] ----
        // Scores stay raw; alpha * log2(e) is folded into the exponent (one fma + v_exp_f32 per score, alpha > 0),
        // and the visibility test is compiled out for tiles below the diagonal (wave-uniform).
        u32x4_t pf[MT][2];
        const bool tile_full = k0 + PF_KEYS <= a.seq_k && (!a.causal || k0 + PF_KEYS - 1 <= q0 + shift);
        if (!tile_full) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int qi = q0 + mt * 16 + ni;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int key = k0 + nt * 16 + kb * 4 + r;
                const bool vis = key < a.seq_k && (!a.causal || key <= qi + shift);
                sacc[mt][nt][r] = vis ? sacc[mt][nt][r] : -INFINITY;
              }
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float mx = sacc[mt][0][0];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[mt][nt][r]);
          mx = rows_max(mx);
          // log2 units; the running maximum starts at -1e30 (finite), so mn is finite, exp2(-inf) = 0 for masked scores
          // and corr = 0 on the first tile without any special case
          const float mn = fmaxf(mrow[mt], mx * sl2);
          const float corr = __builtin_amdgcn_exp2f(mrow[mt] - mn);
          mrow[mt] = mn;
          float psum = 0.f;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              e[r] = __builtin_amdgcn_exp2f(fmaf(sacc[mt][nt][r], sl2, -mn));
              psum += e[r];  // row sum in f32 (P itself is rounded to FT for the matrix core)
            }
            // k-slot j of fragment hh <-> key hh*32 + (j>>2)*16 + kb*4 + (j&3): the order the V fragments are read in
            pf[mt][nt >> 1][(nt & 1) * 2] = pack_ft2<FT>(e[0], e[1]);
            pf[mt][nt >> 1][(nt & 1) * 2 + 1] = pack_ft2<FT>(e[2], e[3]);
          }
          lrow[mt] = lrow[mt] * corr + psum;
          // rescale only when some query's maximum moved (wave-uniform): rare after the first tiles
          if (__builtin_amdgcn_ballot_w64(corr != 1.f) != 0ull) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) oacc[mt][t][r] *= corr;
          }
        }
        // ---- O^T += V^T.P^T: A fragment of dim tile t = V[the 32 keys of fragment hh][t*16 + ni], read transposed from the
        //      row-major tile; shared by the MT query tiles.  Dim tiles 4..7 are read while tiles 0..3 multiply. ----
#pragma unroll
        for (int t = 4; t < 8; ++t) read_v(t);
        if (a.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) oacc[mt][t] = mfma16<FT>(vf[t][hh], pf[mt][hh], oacc[mt][t]);
        __builtin_amdgcn_s_setprio(0);
      } else {
        load_tile(k0 + 2 * kstep);
      }
      __syncthreads();  // tile consumed by every wave; the copy written above becomes visible
    }
    if constexpr (NG == 2) {
      // ---- merge the two key groups: group 1 -> LDS (the K tiles are dead after the loop's last barrier), group 0 adds ----
      f32x4_t* ox = reinterpret_cast<f32x4_t*>(&ks_all[0][0][0]);   // [wave][mt][t][lane]: 4 x MT x 8 KiB <= 64 KiB
      float* mx = reinterpret_cast<float*>(&vs_all[0][0][0]);       // [wave][mt][m, l][lane]
      if (grp == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int t = 0; t < 8; ++t) ox[((wave * MT + mt) * 8 + t) * 64 + lane] = oacc[mt][t];
          mx[((wave * MT + mt) * 2 + 0) * 64 + lane] = mrow[mt];
          mx[((wave * MT + mt) * 2 + 1) * 64 + lane] = lrow[mt];
        }
      }
      __syncthreads();
      if (grp == 1) return;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float m1 = mx[((wave * MT + mt) * 2 + 0) * 64 + lane], l1 = mx[((wave * MT + mt) * 2 + 1) * 64 + lane];
        const float mn = fmaxf(mrow[mt], m1);
        const float c0 = __builtin_amdgcn_exp2f(mrow[mt] - mn), c1 = __builtin_amdgcn_exp2f(m1 - mn);  // a group without tiles: m = -1e30, c = 0 or 1
        lrow[mt] = lrow[mt] * c0 + l1 * c1;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const f32x4_t o1 = ox[((wave * MT + mt) * 8 + t) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[mt][t][r] = oacc[mt][t][r] * c0 + o1[r] * c1;
        }
      }
    }
    // ---- normalise and store: lane holds O[q0 + mt*16 + ni][t*16 + kb*4 .. +4] ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float l = rows_sum(lrow[mt]);
      const int qi = q0 + mt * 16 + ni;
      if (qi < a.seq_q) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        uint16_t* orow = reinterpret_cast<uint16_t*>(a.out) + (size_t)qi * a.n_heads * H + (size_t)head * H + kb * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const uint32_t lo = f32_to_ft_bits<FT>(oacc[mt][t][0] * inv) | (f32_to_ft_bits<FT>(oacc[mt][t][1] * inv) << 16);
          const uint32_t hi = f32_to_ft_bits<FT>(oacc[mt][t][2] * inv) | (f32_to_ft_bits<FT>(oacc[mt][t][3] * inv) << 16);
          *reinterpret_cast<u32x2_t*>(orow + t * 16) = u32x2_t{lo, hi};
        }
      }
    }
  }
}

}  // namespace dihip

using namespace dihip;

extern "C" int dihip_prefill_attn(void* stream, void* out, const void* q, const void* k, const void* v, int seq_q,
                                  int seq_k, int q_stride, int kv_stride, int n_heads, int n_groups, int head_size,
                                  int causal, float alpha, int dtype) {
  DIHIP_REQUIRE(seq_q >= 0 && seq_k >= 0 && n_heads > 0 && n_groups > 0, DIHIP_PARAM_ERROR, "prefill_attn: invalid parameter");
  DIHIP_REQUIRE(head_size == 128, DIHIP_PARAM_ERROR, "prefill_attn: unsupported head size %d (only 128)", head_size);
  DIHIP_REQUIRE(n_heads % n_groups == 0, DIHIP_PARAM_ERROR, "prefill_attn: nHeads must be a multiple of nGroups");
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "prefill_attn: FLOAT16 / BFLOAT16 only");
  DIHIP_REQUIRE(seq_k >= seq_q || !causal, DIHIP_PARAM_ERROR, "prefill_attn: causal attention needs seq_k >= seq_q");
  DIHIP_REQUIRE(alpha > 0.f, DIHIP_PARAM_ERROR, "prefill_attn: the softmax scale must be positive");
  DIHIP_REQUIRE((size_t)(seq_k + 6 * PF_KEYS + 64) * (size_t)(kv_stride > 0 ? kv_stride : 0) * 2 < (1ull << 31), DIHIP_EXCEED_LIMIT_ERROR,
                "prefill_attn: K / V of one call must stay below 2 GiB (32-bit buffer offsets)");
  if (seq_q == 0) return DIHIP_SUCCESS;
  DIHIP_REQUIRE(out && q && k && v && seq_k > 0, DIHIP_PARAM_ERROR, "prefill_attn: null pointer / empty keys");
  DIHIP_REQUIRE(q_stride % 8 == 0 && kv_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(k) & 15) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0,
                DIHIP_PARAM_ERROR, "prefill_attn: rows must be 16-byte aligned");
  static const int force_mt = env_int("DIHIP_PREFILL_MT", 0);  // diagnostics
  int ncu = cached_num_cus();
  if (ncu <= 0) ncu = 256;
  // 128-row query tiles read every K / V fragment once per two MFMAs; 64-row tiles double the workgroup count.
  // Two workgroups fit a CU: take the small tile while the large one would fill less than 1.5 slots (measured).
  auto wgs = [&](int rows) { return (long)((seq_q + rows - 1) / rows) * n_heads; };
  static const int force_ng = env_int("DIHIP_PREFILL_NG", 0);  // diagnostics
  // While all 128-row workgroups are resident at once (2 per CU) the longest one sets the time: two key groups per
  // workgroup halve its chain of key tiles (same 8 waves per CU).  Measured: +12 % at 2048 tokens (448 workgroups);
  // below one workgroup per CU the 64-row tiles are faster still, above two per CU the greedy dispatch balances already
  // (-10..14 % with two groups at 4096 / 8192).
  const int ng = force_ng > 0 ? std::min(force_ng, 2) : (force_mt == 0 && wgs(128) > (long)ncu && wgs(128) <= 2L * ncu ? 2 : 1);
  const int mt = ng == 2 ? 2 : force_mt > 0 ? std::min(force_mt, 2) : (2 * wgs(128) >= 3L * ncu ? 2 : 1);
  const int rows = 64 * mt;
  const int nqt = (seq_q + rows - 1) / rows;
  static const int force_prio = env_int("DIHIP_PREFILL_PRIO", -1);  // diagnostics
  // raised issue priority for the MFMA phases: +9..11 % once the grid exceeds the resident workgroups (2 per CU), +2 % at
  // 16384, but -4 % while every workgroup is resident from the start (measured, A/B in one process)
  const int prio = force_prio >= 0 ? force_prio : ((long)nqt * n_heads > (ng == 2 ? 1L : 2L) * ncu ? 1 : 0);
  PrefillArgs a{out, q, k, v, seq_q, seq_k, q_stride, kv_stride, n_heads, n_groups, causal, alpha, prio};
  const dim3 grid(nqt * n_heads);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define PF_GO(FT_, MT_, NG_) \
  if (dtype == FT_ && mt == MT_ && ng == NG_) hipLaunchKernelGGL((prefill_attn_kernel<FT_, MT_, NG_>), grid, dim3(PF_THREADS * NG_), 0, s, a);
  PF_GO(DIHIP_BF16, 2, 1) PF_GO(DIHIP_BF16, 1, 1) PF_GO(DIHIP_BF16, 2, 2)
  PF_GO(DIHIP_F16, 2, 1) PF_GO(DIHIP_F16, 1, 1) PF_GO(DIHIP_F16, 2, 2)
#undef PF_GO
  return launch_status();
}
