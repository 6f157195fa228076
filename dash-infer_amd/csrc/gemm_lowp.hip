// gemm_lowp.hip -- C-ABI entry points, weight re-layout and launch heuristics of the
// weight-only GEMM (include/dashinfer_hip.h section 1).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "gemm_lowp_launch.hpp"
#include "gemv_stream_kernel.hpp"
#include "gemm_prefill_kernel.hpp"
#include "gemv_batch_kernel.hpp"
#include "gemm_kslice_kernel.hpp"
#include "gemm_panel_kernel.hpp"

namespace dihip {

// ------------------------------------------------------------------------------------------
// "dihip tile-major" weight layout
//   W4: Kp = roundup(K,128), Np = roundup(N,16); 16-byte chunk index ((nt*KT + kt)*64 + lane),
//       lane = kb*16 + n%16, kb = (k%32)/8; dword ks = (k%128)/32 of the chunk holds the 8
//       nibbles j = k%8 at bit 4*(j/2) + 16*(j%2).
//   W8: Kp = roundup(K,64); chunk byte ks*8 + j = u8(q + 128), ks = (k%64)/32.
//   padding (k >= K or n >= N) is zero.
// (scale, zero): uint32 [NTILES][Gp][16] (column-tile major: the parameters a wave needs next to
//   a weight chunk are 64 contiguous bytes, consecutive groups of a tile are adjacent), lo16 =
//   scale bits, hi16 = zero bits (FT), zero padded; Gp = max(G, ceil(Kp / group)).
// ------------------------------------------------------------------------------------------
static inline int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct LowpDims {
  int KTILE, Kp, KT, Np, NTILES, G, Gp, group;
};
static LowpDims lowp_dims(int wbits, int N, int K, int group_size) {
  LowpDims d;
  d.KTILE = wbits == 4 ? 128 : wbits == 8 ? 64 : 32;
  d.Kp = roundup(K, d.KTILE);
  d.KT = d.Kp / d.KTILE;
  d.Np = roundup(N, 16);
  d.NTILES = d.Np / 16;
  d.group = group_size > 0 ? group_size : 0;
  d.G = d.group ? (K + d.group - 1) / d.group : 1;
  d.Gp = d.group ? std::max(d.G, (d.Kp + d.group - 1) / d.group) : 1;
  return d;
}

__global__ void pack_w4_kernel(const uint8_t* __restrict__ wq, uint32_t* __restrict__ out, int N,
                               int K, int KT, int NTILES) {
  const size_t total = (size_t)NTILES * KT * 64 * 4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int ks = idx & 3;
    const int lane = (idx >> 2) & 63;
    const size_t tk = idx >> 8;
    const int kt = (int)(tk % KT), nt = (int)(tk / KT);
    const int n = nt * 16 + (lane & 15), kb = lane >> 4;
    const int NP = (N + 1) / 2;
    uint32_t d = 0;
    if (n < N) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = kt * 128 + ks * 32 + kb * 8 + j;
        if (k < K) {
          const uint8_t b = wq[(size_t)k * NP + (n >> 1)];
          const uint32_t q = (n & 1) ? (b >> 4) : (b & 0xF);  // convert_4bit.h:9-16
          d |= q << (4 * (j >> 1) + 16 * (j & 1));
        }
      }
    }
    out[idx] = d;
  }
}

__global__ void pack_w8_kernel(const int8_t* __restrict__ wq, uint32_t* __restrict__ out, int N,
                               int K, int KT, int NTILES) {
  const size_t total = (size_t)NTILES * KT * 64 * 4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int dw = idx & 3;
    const int lane = (idx >> 2) & 63;
    const size_t tk = idx >> 8;
    const int kt = (int)(tk % KT), nt = (int)(tk / KT);
    const int n = nt * 16 + (lane & 15), kb = lane >> 4;
    const int ks = dw >> 1, j0 = (dw & 1) * 4;
    uint32_t d = 0;
    if (n < N) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kt * 64 + ks * 32 + kb * 8 + j0 + j;
        if (k < K) d |= (uint32_t)(uint8_t)((int)wq[(size_t)k * N + n] + 128) << (8 * j);
      }
    }
    out[idx] = d;
  }
}

// W16: chunk = the 8 consecutive k (FT elements) of column n for k-step kt, lane = kb*16 + n%16
__global__ void pack_w16_kernel(const uint16_t* __restrict__ w, uint32_t* __restrict__ out, int N, int K, int KT,
                                int NTILES) {
  const size_t total = (size_t)NTILES * KT * 64 * 4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int dw = idx & 3;
    const int lane = (idx >> 2) & 63;
    const size_t tk = idx >> 8;
    const int kt = (int)(tk % KT), nt = (int)(tk / KT);
    const int n = nt * 16 + (lane & 15), kb = lane >> 4;
    const int k = kt * 32 + kb * 8 + dw * 2;
    uint32_t d = 0;
    if (n < N) {
      if (k < K) d |= (uint32_t)w[(size_t)k * N + n];
      if (k + 1 < K) d |= (uint32_t)w[(size_t)(k + 1) * N + n] << 16;
    }
    out[idx] = d;
  }
}

__global__ void pack_sz_kernel(const uint16_t* __restrict__ s, const uint16_t* __restrict__ z,
                               uint32_t* __restrict__ out, int N, int Np, int G, int Gp) {
  const size_t total = (size_t)Gp * Np;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(idx & 15);
    const int g = (int)((idx >> 4) % Gp);
    const int n = (int)((idx >> 4) / Gp) * 16 + col;
    uint32_t v = 0;
    if (n < N && g < G) v = (uint32_t)s[(size_t)g * N + n] | ((uint32_t)z[(size_t)g * N + n] << 16);
    out[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------
// launch plan
// ------------------------------------------------------------------------------------------
struct GemmPlan {
  int MT, NT, NTW, col_blocks, m_blocks, splitk, ktiles_per_split, kslice_tiles;
  size_t lds_bytes, slab_bytes;
};

static GemmPlan make_plan(int wbits, int M, int N, int K, int group_size, bool dual) {
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  GemmPlan p;
  p.MT = M > 16 ? 2 : 1;
  const int rows_per_block = 16 * p.MT;
  p.m_blocks = (M + rows_per_block - 1) / rows_per_block;
  const int num_cus = cached_num_cus();
  const int target_blocks = num_cus > 0 ? (num_cus * 5) / 2 : 640;
  // tiles per wave: 4 streams/wave normally; 2 when the matrix is too narrow to fill the chip
  p.NT = 4;
  if (!dual && wbits != 16) {
    const int cb4 = (d.NTILES + GEMM_WAVES * 4 - 1) / (GEMM_WAVES * 4);
    if ((long)cb4 * p.m_blocks * d.KT < target_blocks) p.NT = 2;
  }
  p.NTW = dual ? p.NT / 2 : p.NT;
  p.col_blocks = (d.NTILES + GEMM_WAVES * p.NTW - 1) / (GEMM_WAVES * p.NTW);
  // split-K granule: whole quantisation groups, whole k-tiles
  int granule = 1;
  if (d.group > d.KTILE) granule = (d.group + d.KTILE - 1) / d.KTILE;
  const int units = (d.KT + granule - 1) / granule;
  int want = (int)std::max<long>(1, target_blocks / std::max<long>(1, (long)p.col_blocks * p.m_blocks));
  want = std::min(want, units);
  // a skinny unquantised matrix (the MoE router, hidden -> 64 experts; the shared expert's gate, hidden -> 1) is ONE column block:
  // its last-arriving workgroup sums every split's partial tile alone -- 112 splits of one k-tile took 19.8 us (profiles/r03u),
  // most of it that serial sum
  if (wbits == 16 && (long)p.col_blocks * p.m_blocks <= 2) want = std::min(want, 16);
  int units_per_split = (units + want - 1) / want;
  p.ktiles_per_split = units_per_split * granule;
  p.splitk = (d.KT + p.ktiles_per_split - 1) / p.ktiles_per_split;
  // the activation slice is staged in LDS in pieces of at most ~60 KiB:
  // (rows+1) * (kslice*KTILE + 8) * 2 bytes
  const int rows = std::min(M, rows_per_block);
  const int max_tiles = std::max(1, (60 * 1024 / ((rows + 1) * 2) - 8) / d.KTILE);
  // several slices: each a multiple of the register ring depth (gemm_lowp_kernel.hpp)
  p.kslice_tiles = p.ktiles_per_split <= max_tiles ? p.ktiles_per_split : std::max(GEMM_RING, max_tiles / GEMM_RING * GEMM_RING);
  p.lds_bytes = GEMM_LDS_HEADER + (size_t)(rows + 1) * (p.kslice_tiles * d.KTILE + 8) * 2;
  p.slab_bytes = p.splitk > 1 ? (size_t)p.splitk * M * (dual ? 2 : 1) * d.Np * sizeof(float) : 0;
  return p;
}

constexpr size_t GEMM_SYNC_BYTES = 64 * 1024;  // up to 16384 (column-block x m-block) counters

template <int WBITS, int FT>
static hipError_t dispatch(const GemmPlan& p, int pro, int epi, const GemmArgs& a, dim3 grid,
                           hipStream_t s) {
#define CASE(MT_, NT_, PRO_, EPI_)                                                 \
  if (p.MT == MT_ && p.NT == NT_ && pro == PRO_ && epi == EPI_)                    \
    return launch_gemm_lowp<WBITS, FT, MT_, NT_, PRO_, EPI_>(a, grid, p.lds_bytes, s);
  CASE(1, 2, PRO_PLAIN, EPI_STD)
  CASE(1, 4, PRO_PLAIN, EPI_STD)
  CASE(2, 2, PRO_PLAIN, EPI_STD)
  CASE(2, 4, PRO_PLAIN, EPI_STD)
  {  // fused decode-step forms: bf16 and (round 4) f16
    CASE(1, 2, PRO_RMSNORM, EPI_STD)
    CASE(1, 4, PRO_RMSNORM, EPI_STD)
    CASE(1, 4, PRO_RMSNORM, EPI_SWIGLU)
    CASE(1, 4, PRO_PLAIN, EPI_SWIGLU)
    CASE(2, 4, PRO_PLAIN, EPI_SWIGLU)
    CASE(1, 2, PRO_PLAIN, EPI_ADDTO)
    CASE(1, 4, PRO_PLAIN, EPI_ADDTO)
    CASE(2, 2, PRO_PLAIN, EPI_ADDTO)
    CASE(2, 4, PRO_PLAIN, EPI_ADDTO)
  }
#undef CASE
  return hipErrorInvalidValue;
}

template <int FT>
static hipError_t dispatch_dense(const GemmPlan& p, int pro, int epi, const GemmArgs& a, dim3 grid, hipStream_t s) {
#define CASE(MT_, PRO_, EPI_)                                   \
  if (p.MT == MT_ && p.NT == 4 && pro == PRO_ && epi == EPI_)   \
    return launch_gemm_lowp<16, FT, MT_, 4, PRO_, EPI_>(a, grid, p.lds_bytes, s);
  CASE(1, PRO_RMSNORM, EPI_ADDTO)
  CASE(1, PRO_PLAIN, EPI_ADDTO)
  CASE(2, PRO_PLAIN, EPI_ADDTO)
  CASE(1, PRO_PLAIN, EPI_STD)
  CASE(2, PRO_PLAIN, EPI_STD)
#undef CASE
  return hipErrorInvalidValue;
}

struct GemmCall {
  int wbits, dtype, pro, epi;
  const void* x;
  int ldx;
  const void* gamma;
  float eps;
  const void *w0, *sz0, *w1, *sz1;
  const void *bias, *residual;
  void* y;
  const float* h_res;
  float* h_out;
  int M, N, K, group_size, act;
  float alpha;
  void* ws;
  size_t ws_bytes;
  void* sync;
  int x_layout, y_layout;  // DIHIP_ACT_ROWMAJOR / DIHIP_ACT_FRAG32 (small-batch kernel only)
  // EPI_ADDTO: RMSNorm of the finished rows wanted in n_out (dihip_fused_gemm_addto_norm); *n_done is set when the kernel
  // family that served the call has produced it (split-K slab reduction), otherwise the caller runs the norm kernel
  const void* n_gamma;
  float n_eps;
  void* n_out;
  int n_frag_mt;
  bool* n_done;
  // deferred RMSNorm (gemv_batch_kernel.hpp, GembArgs).  Producer: n_rowsq != null asks for n_out = FT(gamma * h_out) and the
  // row partials instead of the finished norm; *n_parts = parts written, 0 when the serving kernel does not offer it (the
  // norm is then produced as above).  Consumer: rowsq != null -- rows are scaled by 1 / rms on the accumulator.
  float* n_rowsq;
  size_t n_rowsq_bytes;
  int* n_parts;
  const float* rowsq;
  int rowsq_parts;
  float rowsq_eps;
};


// ---- decode fast path (gemv_stream_kernel.hpp) ------------------------------------------------
struct GemvPlan {
  bool ok;
  int MR, upb, blocks, WK, WN, ktpg, kgroups, RS;
  size_t lds_bytes;
};

// K-slice boundaries of a plan (GemvArgs::kcut) and the wave mapping.  Even split: kcut[i] = kgroups * i / WK (the former
// in-kernel formula).  DIHIP_GEMV_KSKEW = s (per cent): the k-slices held by the first-dispatched half of the workgroup's waves
// get (1 + s/100) shares, the others (1 - s/100) -- the two waves of a SIMD are arbitrated by age and the younger half streams
// ~13 % slower (profiles/r03_gemv_wave_timeline.txt); boundaries rounded to whole split units, every slice keeps >= 1 unit when
// there are enough.  DIHIP_GEMV_WMAP = 1: consecutive waves differ in wn, so that the k-slices ARE ordered by age also for the
// WK 4 x WN 2 plans (gate / up pair: without it waves 0-3 stream the gate matrix and 4-7 the up matrix, and no K split can
// balance them).
static void fill_kcut(GemvArgs& g) {
  static const int skew = std::max(0, std::min(60, env_int("DIHIP_GEMV_KSKEW", 0)));
  static const int wmap_env = env_int("DIHIP_GEMV_WMAP", 0);
  g.wmap = (wmap_env && g.WN > 1 && g.WK > 1) ? 1 : 0;
  const int WK = g.WK, G = g.kgroups;
  // which k-slices sit on the older half of the waves
  auto older = [&](int wk) {
    if (WK == 1) return true;
    if (g.WN == 1) return wk < WK / 2;           // wave == wk
    return g.wmap ? wk < WK / 2 : true;           // wmap 0 with WN > 1: every k-slice has waves of both halves
  };
  const bool skewed = skew > 0 && (g.WN == 1 || g.wmap) && WK >= 2 && G >= 2 * WK;
  double tot = 0;
  double w[GEMV_WAVES];
  for (int i = 0; i < WK; ++i) {
    w[i] = skewed ? (older(i) ? 100.0 + skew : 100.0 - skew) : 100.0;
    tot += w[i];
  }
  g.kcut[0] = 0;
  double acc = 0;
  for (int i = 0; i < WK; ++i) {
    acc += w[i];
    int c = skewed ? (int)(G * acc / tot + 0.5) : (int)(((long)G * (i + 1)) / WK);
    c = std::max(c, g.kcut[i]);
    g.kcut[i + 1] = std::min(c, G);
  }
  g.kcut[WK] = G;
  for (int i = WK + 1; i <= GEMV_WAVES; ++i) g.kcut[i] = G;
}

// want_blocks > 0 (expert slots: gridDim.y multiplies the grid): aim at that many workgroups instead of one per CU
static GemvPlan make_gemv_plan(int wbits, int M, int N, int K, int group_size, bool dual, int want_blocks = 0) {
  GemvPlan p{};
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  p.ok = false;
  // DIHIP_GEMV_STREAM_MAXM: largest batch served by the LDS-resident kernel (beyond 4 rows the per-workgroup x staging, M x K
  // into LDS, costs more than the small-batch kernel)
  static const int max_m = std::max(1, std::min(16, env_int("DIHIP_GEMV_STREAM_MAXM", 4)));
  if (M < 1 || M > max_m) return p;
  if (d.group && d.group % d.KTILE != 0) return p;  // groups smaller than a k-tile: general kernel
  p.ktpg = d.group ? d.group / d.KTILE : (1 << 28);
  p.MR = M == 1 ? 1 : 4;
  p.RS = d.Kp + 8;
  int num_cus = cached_num_cus();
  if (num_cus <= 0) num_cus = 256;
  const int units = d.NTILES;
  // never more workgroups than CUs (a second round starts ~2 us late, measured on the qkv shape):
  // one tile per workgroup up to the CU count, ceil(tiles / CUs) beyond
  p.upb = units <= num_cus ? 1 : (units + num_cus - 1) / num_cus;
  static const int upb_override = env_int("DIHIP_GEMV_UPB", 0);  // experiments: units per workgroup for the multi-unit shapes
  if (upb_override > 0 && p.upb > 1) p.upb = upb_override;
  static const int upb_small = env_int("DIHIP_GEMV_UPB_SMALL", 0);  // experiments: units per workgroup for shapes with units in (CUs, 1.5 CUs]
  if (upb_small > 0 && p.upb == 1 && units > num_cus) p.upb = upb_small;
  if (want_blocks > 0) p.upb = std::max(1, (units + want_blocks - 1) / want_blocks);
  p.blocks = (units + p.upb - 1) / p.upb;
  const int nv = p.upb * (dual ? 2 : 1);
  const int kgroups = d.group ? (d.KT + p.ktpg - 1) / p.ktpg : d.KT;
  // wave grid WK x WN: minimise the longest per-wave chunk sequence (ties: fewer k-slices)
  long best = -1;
  p.kgroups = kgroups;
  for (int wn = GEMV_WAVES; wn >= (dual ? 2 : 1); wn /= 2) {
    const int wkk = GEMV_WAVES / wn;
    if (wkk > kgroups) continue;
    const long load = (long)((nv + wn - 1) / wn) * ((kgroups + wkk - 1) / wkk);
    if (best < 0 || load < best) {
      best = load;
      p.WN = wn;
      p.WK = wkk;
    }
  }
  if (best < 0) {
    p.WN = GEMV_WAVES;
    p.WK = 1;
  }
  p.lds_bytes = gemv_lds_bytes(M, p.RS, d.KT, p.upb, dual, p.WK);
  if (p.lds_bytes > 150 * 1024) return p;
  p.ok = true;
  return p;
}

// The decode GEMV's plan for ONE row as the workgroups of ANOTHER launch run it (decode_attn_block.hip): the K split (WK, kcut)
// and the wave grid are those of the stand-alone launch -- the sums are then bit-identical to it -- while the column tiles are dealt
// over `nblocks` workgroups (block b owns tiles b, b + nblocks, ...).  Fills the shape / plan fields of *g (pointers are the
// caller's); *max_units = most tiles one workgroup owns, *lds_bytes = its LDS need.  False: shape not served by the decode GEMV.
bool gemv_block_plan(int wbits, int N, int K, int group_size, int nblocks, GemvArgs* g, int* max_units, size_t* lds_bytes) {
  const GemvPlan gp = make_gemv_plan(wbits, 1, N, K, group_size, false);
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  if (!gp.ok || gp.MR != 1 || K != d.Kp || nblocks < 1 || nblocks > d.NTILES) return false;
  g->M = 1;
  g->N = N;
  g->K = K;
  g->ldx = K;
  g->ldy = N;
  g->KT = d.KT;
  g->NTILES = d.NTILES;
  g->Gp = lowp_dims(4, N, K, group_size).Gp;
  g->ktpg = gp.ktpg;
  g->kgroups = gp.kgroups;
  g->WK = gp.WK;
  g->WN = gp.WN;
  g->RS = gp.RS;
  g->nu_q = d.NTILES / nblocks;
  g->nu_r = d.NTILES % nblocks;
  g->upb = g->nu_q + (g->nu_r ? 1 : 0);
  g->alpha = 1.f;
  g->act = DIHIP_ACT_NONE;
  fill_kcut(*g);
  *max_units = g->upb;
  *lds_bytes = gemv_lds_bytes(1, gp.RS, d.KT, g->upb, 0, gp.WK);
  return true;
}

// The stand-alone decode GEMV's plan and shape fields for one row, as run_gemm fills them (decode_mlp_block.hip runs the bodies of
// two such launches inside one: same blocks, same K split, same sums).  Pointers are the caller's.
bool gemv_plan_args(int wbits, int N, int K, int group_size, bool dual, GemvArgs* g, int* blocks, size_t* lds_bytes) {
  const GemvPlan gp = make_gemv_plan(wbits, 1, N, K, group_size, dual);
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  if (!gp.ok || gp.MR != 1 || K != d.Kp) return false;
  g->M = 1;
  g->N = N;
  g->K = K;
  g->ldx = K;
  g->ldy = N;
  g->KT = d.KT;
  g->NTILES = d.NTILES;
  g->Gp = lowp_dims(4, N, K, group_size).Gp;
  g->ktpg = gp.ktpg;
  g->kgroups = gp.kgroups;
  g->upb = gp.upb;
  g->nu_q = d.NTILES / gp.blocks;
  g->nu_r = d.NTILES % gp.blocks;
  g->WK = gp.WK;
  g->WN = gp.WN;
  g->RS = gp.RS;
  g->alpha = 1.f;
  g->act = DIHIP_ACT_NONE;
  fill_kcut(*g);
  *blocks = gp.blocks;
  *lds_bytes = gp.lds_bytes;
  return true;
}

template <int WBITS, int FT>
static hipError_t dispatch_gemv(const GemvPlan& p, int pro, int epi, const GemvArgs& a, hipStream_t s) {
  const bool gpt = WBITS != 16 && p.ktpg == 1;
#define CASE(MR_, PRO_, EPI_)                                                              \
  if (p.MR == MR_ && pro == PRO_ && epi == EPI_) {                                         \
    if constexpr (WBITS != 16) {                                                           \
      if (gpt) return launch_gemv_stream<WBITS, FT, MR_, PRO_, EPI_, 1>(a, p.blocks, p.lds_bytes, s); \
    }                                                                                      \
    return launch_gemv_stream<WBITS, FT, MR_, PRO_, EPI_, 0>(a, p.blocks, p.lds_bytes, s); \
  }
  CASE(1, PRO_PLAIN, EPI_STD)
  CASE(4, PRO_PLAIN, EPI_STD)
  CASE(1, PRO_RMSNORM, EPI_STD)
  CASE(4, PRO_RMSNORM, EPI_STD)
  CASE(1, PRO_RMSNORM, EPI_SWIGLU)
  CASE(4, PRO_RMSNORM, EPI_SWIGLU)
  CASE(1, PRO_PLAIN, EPI_SWIGLU)
  CASE(4, PRO_PLAIN, EPI_SWIGLU)
  CASE(1, PRO_PLAIN, EPI_ADDTO)
  CASE(4, PRO_PLAIN, EPI_ADDTO)
  CASE(1, PRO_RMSNORM, EPI_ADDTO)
  CASE(4, PRO_RMSNORM, EPI_ADDTO)
#undef CASE
  return hipErrorInvalidValue;
}

// (round 3: f16 activations had the op-boundary form only; round 4 instantiates every form for f16 -- dispatch_gemv<WBITS, DIHIP_F16>)
template <int WBITS>
[[maybe_unused]] static hipError_t dispatch_gemv_f16(const GemvPlan& p, const GemvArgs& a, hipStream_t s) {
  const bool gpt = p.ktpg == 1;
  if (p.MR == 1) return gpt ? launch_gemv_stream<WBITS, DIHIP_F16, 1, PRO_PLAIN, EPI_STD, 1>(a, p.blocks, p.lds_bytes, s)
                            : launch_gemv_stream<WBITS, DIHIP_F16, 1, PRO_PLAIN, EPI_STD, 0>(a, p.blocks, p.lds_bytes, s);
  return gpt ? launch_gemv_stream<WBITS, DIHIP_F16, 4, PRO_PLAIN, EPI_STD, 1>(a, p.blocks, p.lds_bytes, s)
             : launch_gemv_stream<WBITS, DIHIP_F16, 4, PRO_PLAIN, EPI_STD, 0>(a, p.blocks, p.lds_bytes, s);
}

static bool gemv_stream_enabled() {  // DIHIP_GEMV_STREAM=0 routes everything to the general kernel
  static const bool on = !env_off("DIHIP_GEMV_STREAM");
  return on;
}

// ---- batched decode, FRAG32 activations (gemm_panel_kernel.hpp) ----------------------------------------
struct PanelPlan {
  bool ok;
  int panels, nslices, ktps;
  size_t slab_bytes;
};

static PanelPlan make_panel_plan(int wbits, int M, int N, int K, int group_size, bool dual) {
  PanelPlan p{};
  static const bool enabled = !env_off("DIHIP_GEMM_PANEL");  // =0: keep the whole-column kernel (diagnostics)
  if (!enabled) return p;
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  int ncu = cached_num_cus();
  if (ncu <= 0) ncu = 256;
  p.panels = (d.NTILES + PANEL_WAVES - 1) / PANEL_WAVES;
  const int ktpg = d.group ? std::max(1, d.group / d.KTILE) : 1;  // slices hold whole quantisation groups
  static const int target = env_int("DIHIP_PANEL_TARGET_WGS", 0), one_frac = env_int("DIHIP_PANEL_ONE_SLICE_PCT", 50);  // diagnostics
  const int tgt = target > 0 ? target : ncu;
  if (100 * p.panels >= one_frac * ncu) {
    p.nslices = 1;  // enough panels to keep the HBM queue full from every second CU on
    p.ktps = d.KT;
  } else {
    // few columns: split K so that about one workgroup lands on every CU, but keep >= 8 k-tiles per slice
    int s = std::max(1, tgt / p.panels);
    int ktps = (d.KT + s - 1) / s;
    ktps = (ktps + ktpg - 1) / ktpg * ktpg;
    if (ktps < 8) return p;  // the ring would never fill: the whole-column kernel handles these
    p.ktps = ktps;
    p.nslices = (d.KT + ktps - 1) / ktps;
  }
  p.slab_bytes = p.nslices > 1 ? (size_t)p.nslices * (dual ? 2 : 1) * M * N * sizeof(float) : 0;
  p.ok = true;
  return p;
}

// ---- batched decode, FRAG32 activations, register-resident K-slices (gemm_kslice_kernel.hpp) ---------------------
// Context-phase GEMM, tail split (gemm_prefill_kernel.hpp, PrefillArgs::tail_cb): when the grid of 128 x 256 tiles ends in a
// round that fills at most half the chip, the column blocks of that round are split in K instead -- `ksplit` workgroups per
// tile, whole quantisation groups each, all of them within one round -- and a reduction launch adds the parts.  Qwen2-7B at
// 2048 rows: qkv 288 tiles -> 256 + 32 x 7 parts (176 -> ~115 us), gate / up 2368 -> 2304 + 64 x 4.  DIHIP_PREFILL_TAIL_SPLIT=0: off.
struct PrefillTail {
  int col_blocks, mblocks, tail_cb, ksplit;
  size_t slab_bytes;
};
static PrefillTail prefill_tail_plan(int wbits, int M, int N, int K, int group_size, bool dual) {
  PrefillTail t{};
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  const int tpb = dual ? PF_WN * PF_CW / 2 : PF_WN * PF_CW;
  t.col_blocks = (d.NTILES + tpb - 1) / tpb;
  t.mblocks = (M + PF_BM - 1) / PF_BM;
  t.tail_cb = t.col_blocks;
  t.ksplit = 1;
  static const bool on = !env_off("DIHIP_PREFILL_TAIL_SPLIT");
  int ncu = cached_num_cus();
  if (ncu <= 0) ncu = 256;
  const int blocks = t.col_blocks * t.mblocks, rem = blocks % ncu;
  if (!on || M < 64 || (wbits != 4 && wbits != 8) || blocks <= ncu || rem == 0 || rem > ncu / 2) return t;
  const int tcb = (rem + t.mblocks - 1) / t.mblocks, tiles = tcb * t.mblocks;
  if (tiles > ncu / 2 || tcb >= t.col_blocks) return t;
  const int ktpg = d.group ? d.group / d.KTILE : d.KT;
  const int units = d.group ? d.KT / std::max(ktpg, 1) : d.KT;  // K parts hold whole groups (per-channel: whole k-tiles)
  int best = 1;
  for (int ks = 2; ks <= 8; ++ks)
    if (units % ks == 0 && tiles * ks <= ncu) best = ks;
  if (best == 1) return t;
  t.tail_cb = t.col_blocks - tcb;
  t.ksplit = best;
  t.slab_bytes = (size_t)tiles * best * PF_BM * 256 * sizeof(float);
  return t;
}

struct KslicePlan {
  bool ok;
  int groups, nslices, waves, nunits;
  size_t slab_bytes;
};
static KslicePlan make_kslice_plan(int wbits, int M, int N, int K, int group_size, bool dual, bool sizing = false) {
  KslicePlan p{};
  // DIHIP_GEMM_KSLICE: 0 = never, 2 = every eligible shape (tests / diagnostics); read per call so that one process can
  // exercise both kernels
  const char* e = getenv("DIHIP_GEMM_KSLICE");
  const int mode = sizing ? 2 : e ? atoi(e) : 1;  // workspace sizing covers the forced mode as well
  if (mode == 0 || (wbits != 4 && wbits != 8)) return p;
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  const int ch = wbits == 4 ? 4 : 8;  // k-tiles per K-slice (16 k-steps of 32)
  if (d.KT % ch) return p;            // slices hold whole k-tiles of equal count
  if (wbits == 8 && d.group == d.KTILE) return p;  // W8 g64 (a scale per k-tile in a 16-slot ring) does not fit the register file
  // Measured against the panel kernel (tools/gemv_bench and bench.py, 7B and 72B/TP8 shapes): +13 % on the SwiGLU pair
  // at M = 16 (8 KiB in flight per wave), +4 % (gate/up) and +8 % (down, split-K slab) at M = 32, but 30-40 % slower on
  // matrices of a few MB (one or two half-units per workgroup: the ring never reaches steady state)
  if (mode != 2 && (size_t)N * K * wbits / 8 * (dual ? 2 : 1) < ((size_t)24 << 20)) return p;
  int ncu = cached_num_cus();
  if (ncu <= 0) ncu = 256;
  const int nsl = d.KT / ch;
  p.waves = std::min(KSL_WAVES, nsl);
  p.nslices = (nsl + KSL_WAVES - 1) / KSL_WAVES;
  p.nunits = d.NTILES;
  static const int target = env_int("DIHIP_KSLICE_TARGET_WGS", 0);  // diagnostics
  const int tgt = target > 0 ? target : ncu;  // one 8-wave workgroup per CU (64 KB LDS, ~250 registers)
  p.groups = std::max(1, std::min(p.nunits, tgt / p.nslices));
  p.slab_bytes = p.nslices > 1 ? (size_t)p.nslices * (dual ? 2 : 1) * M * N * sizeof(float) : 0;
  p.ok = true;
  return p;
}

// One launch of M = 1 GEMVs over (token, expert-rank) slots (mixture-of-experts, moe.hip): slot s streams expert
// slot_expert[s] of a stack of equally shaped packed weights, reads activation row s / x_div, writes row s.
int run_gemv_slots(hipStream_t stream, int wbits, int epi, const void* x, int ldx, int x_div, const void* w0, const void* sz0,
                   const void* w1, const void* sz1, void* y, int N, int K, int group_size, const int* slot_expert, int nslots,
                   const int* group_rows, const int* group_nrows) {
  const bool grouped = group_rows != nullptr;  // slot_expert / nslots then enumerate groups of <= 4 slots (moe.hip)
  const bool dual = epi == EPI_SWIGLU;
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  DIHIP_REQUIRE(K == d.Kp && (d.group == 0 || d.group % d.KTILE == 0), DIHIP_PARAM_ERROR,
                "moe: K = %d must be a multiple of the k-tile (%d) and groups whole k-tiles", K, d.KTILE);
  // slots multiply the grid: about one workgroup per CU over the whole launch, at most 8 units per workgroup
  int ncu = cached_num_cus();
  if (ncu <= 0) ncu = 256;
  const int want = std::max((d.NTILES + 7) / 8, (ncu + nslots - 1) / nslots);
  const GemvPlan gp = make_gemv_plan(wbits, grouped ? 4 : 1, N, K, group_size, dual, want);
  DIHIP_REQUIRE(gp.ok && gp.lds_bytes <= 64 * 1024, DIHIP_PARAM_ERROR, "moe: unsupported expert shape N=%d K=%d", N, K);
  GemvArgs g{};
  g.w0 = reinterpret_cast<const u32x4_t*>(w0);
  g.w1 = reinterpret_cast<const u32x4_t*>(w1);
  g.sz0 = reinterpret_cast<const uint32_t*>(sz0);
  g.sz1 = reinterpret_cast<const uint32_t*>(sz1);
  g.x = x;
  g.ldx = ldx;
  g.y = y;
  g.ldy = N;
  g.alpha = 1.f;
  g.act = DIHIP_ACT_NONE;
  g.M = grouped ? 4 : 1;
  g.slot_rows = group_rows;
  g.slot_nrows = group_nrows;
  g.N = N;
  g.K = K;
  g.KT = d.KT;
  g.NTILES = d.NTILES;
  g.Gp = lowp_dims(4, N, K, group_size).Gp;
  g.ktpg = gp.ktpg;
  g.kgroups = gp.kgroups;
  g.upb = gp.upb;
  g.nu_q = d.NTILES / gp.blocks;
  g.nu_r = d.NTILES % gp.blocks;
  g.WK = gp.WK;
  g.WN = gp.WN;
  g.RS = gp.RS;
  fill_kcut(g);
  g.slot_expert = slot_expert;
  g.w_estride = (size_t)d.NTILES * d.KT * 64;       // u32x4 per expert
  g.sz_estride = (size_t)d.NTILES * g.Gp * 16;      // u32 per expert
  g.x_div = x_div;
  g.nslots = nslots;
  const bool gpt = gp.ktpg == 1;
  hipError_t e = hipErrorInvalidValue;
#define SLOT_GO(W_, EPI_, G_)                                                                                          \
  if (wbits == W_ && epi == EPI_ && (int)gpt == G_)                                                                    \
    e = grouped ? launch_gemv_slots<W_, DIHIP_BF16, EPI_, G_, 4>(g, gp.blocks, gp.lds_bytes, stream)                   \
                : launch_gemv_slots<W_, DIHIP_BF16, EPI_, G_, 1>(g, gp.blocks, gp.lds_bytes, stream);
  SLOT_GO(8, EPI_SWIGLU, 0) SLOT_GO(8, EPI_STD, 0) SLOT_GO(8, EPI_SWIGLU, 1) SLOT_GO(8, EPI_STD, 1)
  SLOT_GO(4, EPI_SWIGLU, 0) SLOT_GO(4, EPI_STD, 0) SLOT_GO(4, EPI_SWIGLU, 1) SLOT_GO(4, EPI_STD, 1)
#undef SLOT_GO
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "moe: expert GEMV launch failed: %s", hipGetErrorString(e));
  return DIHIP_SUCCESS;
}

static int run_gemm(hipStream_t stream, const GemmCall& c) {
  DIHIP_REQUIRE(c.M >= 0 && c.N > 0 && c.K > 0, DIHIP_PARAM_ERROR, "gemm_lowp: bad shape M=%d N=%d K=%d",
                c.M, c.N, c.K);
  if (c.M == 0) return DIHIP_SUCCESS;  // empty batch
  DIHIP_REQUIRE(c.dtype == DIHIP_BF16 || c.dtype == DIHIP_F16, DIHIP_PARAM_ERROR,
                "gemm_lowp: activation type must be FLOAT16 or BFLOAT16 (gemm_a16w8.cpp:21-125)");
  DIHIP_REQUIRE(c.group_size <= 0 || c.group_size % 32 == 0, DIHIP_PARAM_ERROR,
                "gemm_lowp: GroupSize %d must be a multiple of 32", c.group_size);
  DIHIP_REQUIRE(c.x && c.w0 && (c.sz0 || c.wbits == 16), DIHIP_PARAM_ERROR, "gemm_lowp: null input pointer");
  const bool dual = c.epi == EPI_SWIGLU;
  const LowpDims d = lowp_dims(c.wbits, c.N, c.K, c.group_size);
  const bool gemv_aligned = (c.K == d.Kp) && (c.ldx % 8 == 0) && (reinterpret_cast<uintptr_t>(c.x) % 16 == 0) &&
                            (c.pro == PRO_PLAIN || reinterpret_cast<uintptr_t>(c.gamma) % 16 == 0);
  const bool f16_std = c.dtype == DIHIP_F16;  // every form of the decode GEMV exists for f16 as well
  const bool want_frag = c.x_layout == DIHIP_ACT_FRAG32 || c.y_layout == DIHIP_ACT_FRAG32;  // small-batch kernel only
  DIHIP_REQUIRE(!c.rowsq || (c.M > 4 && c.M <= 32 && c.dtype == DIHIP_BF16 && c.rowsq_parts > 0 && c.epi != EPI_ADDTO), DIHIP_PARAM_ERROR,
                "gemm_lowp: deferred row norms are taken by the small-batch kernels only (4 < M <= 32, bf16); see dihip_prenorm_rowsq_supported");
  if (c.n_parts) *c.n_parts = 0;
  if ((c.dtype == DIHIP_BF16 || f16_std) && gemv_stream_enabled() && gemv_aligned && !want_frag) {
    const GemvPlan gp = make_gemv_plan(c.wbits, c.M, c.N, c.K, c.group_size, dual);
    if (gp.ok) {
      GemvArgs g{};
      g.w0 = reinterpret_cast<const u32x4_t*>(c.w0);
      g.w1 = reinterpret_cast<const u32x4_t*>(c.w1);
      g.sz0 = reinterpret_cast<const uint32_t*>(c.sz0);
      g.sz1 = reinterpret_cast<const uint32_t*>(c.sz1);
      g.x = c.x;
      g.ldx = c.ldx;
      g.gamma = c.gamma;
      g.eps = c.eps;
      g.bias = c.bias;
      g.residual = c.residual;
      g.y = c.y;
      g.ldy = c.N;
      g.h_res = c.h_res;
      g.h_out = c.h_out;
      g.alpha = c.alpha;
      g.act = c.act;
      g.M = c.M;
      g.N = c.N;
      g.K = c.K;
      g.KT = d.KT;
      g.NTILES = d.NTILES;
      g.Gp = lowp_dims(4, c.N, c.K, c.group_size).Gp;
      g.ktpg = gp.ktpg;
      g.kgroups = gp.kgroups;
      g.upb = gp.upb;
      g.nu_q = d.NTILES / gp.blocks;
      g.nu_r = d.NTILES % gp.blocks;
      g.WK = gp.WK;
      g.WN = gp.WN;
      g.RS = gp.RS;
      fill_kcut(g);
      g.trace = debug_trace_buffer((size_t)gp.blocks * GEMV_WAVES * 64);
      hipError_t e = hipErrorInvalidValue;
      if (f16_std && c.wbits == 4) e = dispatch_gemv<4, DIHIP_F16>(gp, c.pro, c.epi, g, stream);
      else if (f16_std && c.wbits == 8) e = dispatch_gemv<8, DIHIP_F16>(gp, c.pro, c.epi, g, stream);
      else if (f16_std && c.wbits == 16) e = dispatch_gemv<16, DIHIP_F16>(gp, c.pro, c.epi, g, stream);
      else if (c.wbits == 4) e = dispatch_gemv<4, DIHIP_BF16>(gp, c.pro, c.epi, g, stream);
      else if (c.wbits == 8) e = dispatch_gemv<8, DIHIP_BF16>(gp, c.pro, c.epi, g, stream);
      else if (c.wbits == 16) e = dispatch_gemv<16, DIHIP_BF16>(gp, c.pro, c.epi, g, stream);
      DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "gemv_stream: launch failed (wbits=%d MR=%d pro=%d epi=%d): %s",
                    c.wbits, gp.MR, c.pro, c.epi, hipGetErrorString(e));
      return DIHIP_SUCCESS;
    }
  }
  // batched decode with FRAG32 activations: panels of 8 column tiles sharing x through LDS (gemm_panel_kernel.hpp)
  const bool f16_act = c.dtype == DIHIP_F16;  // (round 5: the small-batch and context-phase kernels are instantiated for f16 as well)
  if ((c.dtype == DIHIP_BF16 || f16_act) && gemv_stream_enabled() && c.x_layout == DIHIP_ACT_FRAG32 && c.pro == PRO_PLAIN && c.M > 4 &&
      c.M <= 32 && c.wbits != 16 && c.K == d.Kp && (d.group == 0 || d.group % d.KTILE == 0)) {
    const KslicePlan kp = make_kslice_plan(c.wbits, c.M, c.N, c.K, c.group_size, dual);
    if (kp.ok) {
      DIHIP_REQUIRE(kp.nslices == 1 || (c.ws && c.ws_bytes >= kp.slab_bytes), DIHIP_MEMORY_ERROR,
                    "gemm_kslice: workspace too small for the split-K slab (%zu < %zu)", c.ws_bytes, kp.slab_bytes);
      PanelArgs g{};
      g.w0 = reinterpret_cast<const u32x4_t*>(c.w0);
      g.w1 = reinterpret_cast<const u32x4_t*>(c.w1);
      g.sz0 = reinterpret_cast<const uint32_t*>(c.sz0);
      g.sz1 = reinterpret_cast<const uint32_t*>(c.sz1);
      g.x = c.x;
      g.bias = c.bias;
      g.residual = c.residual;
      g.y = c.y;
      g.ldy = c.N;
      g.h_res = c.h_res;
      g.h_out = c.h_out;
      g.alpha = c.alpha;
      g.act = c.act;
      g.M = c.M;
      g.N = c.N;
      g.K = c.K;
      g.KT = d.KT;
      g.NTILES = d.NTILES;
      g.Gp = lowp_dims(4, c.N, c.K, c.group_size).Gp;
      g.ktpg = d.group ? d.group / d.KTILE : (1 << 28);
      g.nslices = kp.nslices;
      g.slab = reinterpret_cast<float*>(c.ws);
      g.yfrag = c.y_layout == DIHIP_ACT_FRAG32;
      g.nunits = kp.nunits;
      g.trace = debug_trace_buffer((size_t)kp.groups * kp.nslices * KSL_WAVES * 64);
      if (c.epi == EPI_ADDTO && c.n_gamma && c.n_out && g.nslices > 1 && c.N % 4 == 0 && c.N <= 8192) {
        g.n_gamma = c.n_gamma;
        g.n_eps = c.n_eps;
        g.n_out = c.n_out;
        g.n_frag_mt = c.n_frag_mt;
        if (c.n_done) *c.n_done = true;
      }
      const bool gpt = g.ktpg == 1;
      const int mt = c.M > 16 ? 2 : 1;
      const int waves = kp.waves;
      if (c.rowsq) {  // deferred row norms: in the epilogue (unsplit K) or in the slab reduction
        g.rowsq = c.rowsq;
        g.rowsq_parts = c.rowsq_parts;
        g.rowsq_eps = c.rowsq_eps;
      }
      hipError_t e = hipErrorInvalidValue;
#define KSLICE_GO(W_, MT_, EPI_, G_) \
      if (c.wbits == W_ && mt == MT_ && c.epi == EPI_ && (int)gpt == G_)                                                       \
        e = f16_act ? launch_gemm_kslice<W_, DIHIP_F16, MT_, EPI_, G_>(g, kp.groups, waves, stream)                            \
                    : launch_gemm_kslice<W_, DIHIP_BF16, MT_, EPI_, G_>(g, kp.groups, waves, stream);
#define KSLICE_ALL(W_, G_) KSLICE_GO(W_, 1, EPI_STD, G_) KSLICE_GO(W_, 2, EPI_STD, G_) KSLICE_GO(W_, 1, EPI_SWIGLU, G_) \
      KSLICE_GO(W_, 2, EPI_SWIGLU, G_) KSLICE_GO(W_, 1, EPI_ADDTO, G_) KSLICE_GO(W_, 2, EPI_ADDTO, G_)
      if (c.rowsq && kp.nslices == 1) {  // the deferred-RMSNorm consumer has its own instantiations (bf16)
#define KSLICE_RS_GO(W_, MT_, EPI_, G_) \
        if (c.wbits == W_ && mt == MT_ && c.epi == EPI_ && (int)gpt == G_) e = launch_gemm_kslice_rs<W_, MT_, EPI_, G_>(g, kp.groups, waves, stream);
#define KSLICE_RS_ALL(W_, G_) KSLICE_RS_GO(W_, 1, EPI_STD, G_) KSLICE_RS_GO(W_, 2, EPI_STD, G_) KSLICE_RS_GO(W_, 1, EPI_SWIGLU, G_) KSLICE_RS_GO(W_, 2, EPI_SWIGLU, G_)
        KSLICE_RS_ALL(4, 0) KSLICE_RS_ALL(4, 1) KSLICE_RS_ALL(8, 0)
#undef KSLICE_RS_ALL
#undef KSLICE_RS_GO
      } else {
      KSLICE_ALL(4, 0) KSLICE_ALL(4, 1) KSLICE_ALL(8, 0)  // W8 with a group per k-tile (g64) is excluded by the plan: not instantiated
      }
#undef KSLICE_ALL
#undef KSLICE_GO
      DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "gemm_kslice: launch failed (wbits=%d M=%d epi=%d): %s", c.wbits, c.M,
                    c.epi, hipGetErrorString(e));
      return DIHIP_SUCCESS;
    }
    const PanelPlan pp = make_panel_plan(c.wbits, c.M, c.N, c.K, c.group_size, dual);
    if (pp.ok) {
      DIHIP_REQUIRE(pp.nslices == 1 || (c.ws && c.ws_bytes >= pp.slab_bytes), DIHIP_MEMORY_ERROR,
                    "gemm_panel: workspace too small for the split-K slab (%zu < %zu)", c.ws_bytes, pp.slab_bytes);
      PanelArgs g{};
      g.w0 = reinterpret_cast<const u32x4_t*>(c.w0);
      g.w1 = reinterpret_cast<const u32x4_t*>(c.w1);
      g.sz0 = reinterpret_cast<const uint32_t*>(c.sz0);
      g.sz1 = reinterpret_cast<const uint32_t*>(c.sz1);
      g.x = c.x;
      g.bias = c.bias;
      g.residual = c.residual;
      g.y = c.y;
      g.ldy = c.N;
      g.h_res = c.h_res;
      g.h_out = c.h_out;
      g.alpha = c.alpha;
      g.act = c.act;
      g.M = c.M;
      g.N = c.N;
      g.K = c.K;
      g.KT = d.KT;
      g.NTILES = d.NTILES;
      g.Gp = lowp_dims(4, c.N, c.K, c.group_size).Gp;
      g.ktpg = d.group ? d.group / d.KTILE : (1 << 28);
      g.ktps = pp.ktps;
      g.nslices = pp.nslices;
      g.slab = reinterpret_cast<float*>(c.ws);
      g.yfrag = c.y_layout == DIHIP_ACT_FRAG32;
      if (c.epi == EPI_ADDTO && c.n_gamma && c.n_out && g.nslices > 1 && c.N % 4 == 0 && c.N <= 8192) {
        g.n_gamma = c.n_gamma;
        g.n_eps = c.n_eps;
        g.n_out = c.n_out;
        g.n_frag_mt = c.n_frag_mt;
        if (c.n_done) *c.n_done = true;
      }
      const bool gpt = g.ktpg == 1;
      const int mt = c.M > 16 ? 2 : 1;
      if (c.rowsq) {
        DIHIP_REQUIRE(pp.nslices > 1, DIHIP_PARAM_ERROR, "gemm_panel: deferred row norms need the split-K reduction; see dihip_prenorm_rowsq_supported");
        g.rowsq = c.rowsq;
        g.rowsq_parts = c.rowsq_parts;
        g.rowsq_eps = c.rowsq_eps;
      }
      hipError_t e = hipErrorInvalidValue;
#define PANEL_GO(W_, MT_, EPI_, G_) \
      if (c.wbits == W_ && mt == MT_ && c.epi == EPI_ && (int)gpt == G_)                                          \
        e = f16_act ? launch_gemm_panel<W_, DIHIP_F16, MT_, EPI_, G_>(g, pp.panels, stream)                       \
                    : launch_gemm_panel<W_, DIHIP_BF16, MT_, EPI_, G_>(g, pp.panels, stream);
#define PANEL_ALL(W_, G_) PANEL_GO(W_, 1, EPI_STD, G_) PANEL_GO(W_, 2, EPI_STD, G_) PANEL_GO(W_, 1, EPI_SWIGLU, G_) \
      PANEL_GO(W_, 2, EPI_SWIGLU, G_) PANEL_GO(W_, 1, EPI_ADDTO, G_) PANEL_GO(W_, 2, EPI_ADDTO, G_)
      PANEL_ALL(4, 0) PANEL_ALL(4, 1) PANEL_ALL(8, 0) PANEL_ALL(8, 1)
#undef PANEL_ALL
#undef PANEL_GO
      DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "gemm_panel: launch failed (wbits=%d M=%d epi=%d): %s", c.wbits, c.M,
                    c.epi, hipGetErrorString(e));
      return DIHIP_SUCCESS;
    }
  }
  // small decode batches whose activations do not fit in LDS: register-resident A (gemv_batch_kernel.hpp)
  if ((c.dtype == DIHIP_BF16 || f16_act) && gemv_stream_enabled() && gemv_aligned && c.pro == PRO_PLAIN && c.M > 1 && c.M <= 32 &&
      c.wbits != 16 && (d.group == 0 || d.group % d.KTILE == 0)) {
    GembArgs g{};
    g.w0 = reinterpret_cast<const u32x4_t*>(c.w0);
    g.w1 = reinterpret_cast<const u32x4_t*>(c.w1);
    g.sz0 = reinterpret_cast<const uint32_t*>(c.sz0);
    g.sz1 = reinterpret_cast<const uint32_t*>(c.sz1);
    g.x = c.x;
    g.ldx = c.ldx;
    g.bias = c.bias;
    g.residual = c.residual;
    g.y = c.y;
    g.ldy = c.N;
    g.h_res = c.h_res;
    g.h_out = c.h_out;
    g.alpha = c.alpha;
    g.act = c.act;
    g.M = c.M;
    g.N = c.N;
    g.K = c.K;
    g.KT = d.KT;
    g.NTILES = d.NTILES;
    g.Gp = lowp_dims(4, c.N, c.K, c.group_size).Gp;
    g.ktpg = d.group ? d.group / d.KTILE : (1 << 28);
    g.kgroups = d.group ? (d.KT + g.ktpg - 1) / g.ktpg : d.KT;
    const bool gpt = c.wbits != 16 && g.ktpg == 1;
    const int mt = c.M > 16 ? 2 : 1;
    g.xfrag = c.x_layout == DIHIP_ACT_FRAG32;
    g.yfrag = c.y_layout == DIHIP_ACT_FRAG32;
    // at most one workgroup per CU; a workgroup with several units walks them two at a time so that
    // one pass over the activations (L2 -> registers) feeds two column tiles
    static const int env_upb = env_int("DIHIP_GEMB_UPB", 0), env_nt = env_int("DIHIP_GEMB_NT", 0);  // diagnostics
    int ncu = cached_num_cus();
    if (ncu <= 0) ncu = 256;
    g.upb = env_upb > 0 ? env_upb : (d.NTILES <= ncu ? 1 : (d.NTILES + ncu - 1) / ncu);
    const int nt = env_nt > 0 ? std::min(env_nt, 2) : (g.upb >= 2 ? 2 : 1);
    const int blocks = (d.NTILES + g.upb - 1) / g.upb;
    if (c.epi == EPI_ADDTO && c.n_rowsq && c.n_gamma && c.n_out && c.n_parts && c.dtype == DIHIP_BF16 && g.upb <= GEMB_MAXU &&
        (size_t)blocks * 32 * sizeof(float) <= c.n_rowsq_bytes) {  // deferred RMSNorm, producer side
      g.n_gamma = c.n_gamma;
      g.n_out = c.n_out;
      g.n_frag_mt = c.n_frag_mt;
      g.n_rowsq = c.n_rowsq;
      *c.n_parts = blocks;
      if (c.n_done) *c.n_done = true;
    }
    if (c.rowsq) {
      DIHIP_REQUIRE(c.epi == EPI_STD, DIHIP_PARAM_ERROR, "gemv_batch: deferred row norms on the plain epilogue only; see dihip_prenorm_rowsq_supported");
      DIHIP_REQUIRE(c.rowsq_parts <= 16 * (GEMB_THREADS / 32), DIHIP_PARAM_ERROR, "gemv_batch: more than %d row-norm parts", 16 * (GEMB_THREADS / 32));
      g.rowsq = c.rowsq;
      g.rowsq_parts = c.rowsq_parts;
      g.rowsq_eps = c.rowsq_eps;
    }
    hipError_t e = hipErrorInvalidValue;
#define GEMB_GO(W_, MT_, NT_, EPI_, G_) \
    if (c.wbits == W_ && mt == MT_ && nt == NT_ && c.epi == EPI_ && (int)gpt == G_) \
      e = f16_act ? launch_gemv_batch<W_, DIHIP_F16, MT_, NT_, EPI_, G_>(g, blocks, stream)  \
                  : launch_gemv_batch<W_, DIHIP_BF16, MT_, NT_, EPI_, G_>(g, blocks, stream);
#define GEMB_EPIS(W_, MT_, NT_, G_) GEMB_GO(W_, MT_, NT_, EPI_STD, G_) GEMB_GO(W_, MT_, NT_, EPI_SWIGLU, G_) GEMB_GO(W_, MT_, NT_, EPI_ADDTO, G_)
#define GEMB_ALL(W_, G_) GEMB_EPIS(W_, 1, 1, G_) GEMB_EPIS(W_, 2, 1, G_) GEMB_EPIS(W_, 1, 2, G_) GEMB_EPIS(W_, 2, 2, G_)
    GEMB_ALL(4, 0) GEMB_ALL(4, 1) GEMB_ALL(8, 0) GEMB_ALL(8, 1)
#undef GEMB_ALL
#undef GEMB_EPIS
#undef GEMB_GO
    DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "gemv_batch: launch failed (wbits=%d M=%d epi=%d): %s", c.wbits, c.M, c.epi,
                  hipGetErrorString(e));
    return DIHIP_SUCCESS;
  }
  DIHIP_REQUIRE(!want_frag, DIHIP_PARAM_ERROR,
                "gemm_lowp: the FRAG32 activation layout needs the small-batch kernel (see dihip_gemm_lowp_prefers_frag)");
  DIHIP_REQUIRE(!c.rowsq, DIHIP_PARAM_ERROR, "gemm_lowp: this shape is not served by a kernel that takes deferred row norms");
  // context phase (M >= 64 rows): 128 x 256 workgroup tiles, A through LDS, every weight byte read once per 128 rows
  // (gemm_prefill_kernel.hpp).  DIHIP_GEMM_PREFILL=0 keeps the general kernel (A/B, diagnostics).
  static const bool prefill_on = !env_off("DIHIP_GEMM_PREFILL");
  if (prefill_on && (c.dtype == DIHIP_BF16 || f16_act) && c.pro == PRO_PLAIN && c.M >= 64 && (c.wbits == 4 || c.wbits == 8) && gemv_aligned &&
      (d.group == 0 || d.group % d.KTILE == 0) && (c.epi != EPI_SWIGLU || (c.w1 && c.sz1))) {
    PrefillArgs g{};
    g.w0 = reinterpret_cast<const u32x4_t*>(c.w0);
    g.w1 = reinterpret_cast<const u32x4_t*>(c.w1);
    g.sz0 = reinterpret_cast<const uint32_t*>(c.sz0);
    g.sz1 = reinterpret_cast<const uint32_t*>(c.sz1);
    g.x = c.x;
    g.ldx = c.ldx;
    g.bias = c.bias;
    g.residual = c.residual;
    g.y = c.y;
    g.ldy = c.N;
    g.h_res = c.h_res;
    g.h_out = c.h_out;
    g.alpha = c.alpha;
    g.act = c.act;
    g.M = c.M;
    g.N = c.N;
    g.KT = d.KT;
    g.NTILES = d.NTILES;
    g.Gp = lowp_dims(4, c.N, c.K, c.group_size).Gp;
    g.ktpg = d.group ? d.group / d.KTILE : (1 << 28);
    const int tiles_per_block = dual ? PF_WN * PF_CW / 2 : PF_WN * PF_CW;
    g.col_blocks = (d.NTILES + tiles_per_block - 1) / tiles_per_block;
    g.tail_cb = g.col_blocks;
    g.ksplit = 1;
    int blocks = g.col_blocks * ((c.M + PF_BM - 1) / PF_BM);
    const PrefillTail pt = prefill_tail_plan(c.wbits, c.M, c.N, c.K, c.group_size, dual);
    if (pt.ksplit > 1 && c.ws && c.ws_bytes >= pt.slab_bytes && reinterpret_cast<uintptr_t>(c.ws) % 16 == 0) {
      g.tail_cb = pt.tail_cb;
      g.ksplit = pt.ksplit;
      g.slab = reinterpret_cast<float*>(c.ws);
      blocks = pt.tail_cb * pt.mblocks + (pt.col_blocks - pt.tail_cb) * pt.mblocks * pt.ksplit;
    }
    g.trace = debug_trace_buffer((size_t)blocks * PF_WAVES * 4 * 8 * sizeof(unsigned long long));
    const bool gpt = g.ktpg == 1;
    hipError_t e = hipErrorInvalidValue;
#define PREFILL_GO(W_, EPI_, G_) \
    if (c.wbits == W_ && c.epi == EPI_ && (int)gpt == G_)                                          \
      e = f16_act ? launch_gemm_prefill<W_, DIHIP_F16, EPI_, G_>(g, blocks, stream) : launch_gemm_prefill<W_, DIHIP_BF16, EPI_, G_>(g, blocks, stream);
#define PREFILL_ALL(W_) PREFILL_GO(W_, EPI_STD, 0) PREFILL_GO(W_, EPI_STD, 1) PREFILL_GO(W_, EPI_SWIGLU, 0) PREFILL_GO(W_, EPI_SWIGLU, 1) \
    PREFILL_GO(W_, EPI_ADDTO, 0) PREFILL_GO(W_, EPI_ADDTO, 1)
    PREFILL_ALL(4) PREFILL_ALL(8)
#undef PREFILL_ALL
#undef PREFILL_GO
    DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "gemm_prefill: launch failed (wbits=%d M=%d epi=%d): %s", c.wbits, c.M, c.epi,
                  hipGetErrorString(e));
    return DIHIP_SUCCESS;
  }
  const GemmPlan p = make_plan(c.wbits, c.M, c.N, c.K, c.group_size, dual);
  DIHIP_REQUIRE((size_t)p.col_blocks * p.m_blocks * sizeof(unsigned) <= GEMM_SYNC_BYTES, DIHIP_EXCEED_LIMIT_ERROR,
                "gemm_lowp: too many tiles for the sync buffer");
  unsigned* counters = reinterpret_cast<unsigned*>(c.sync);
  float* slabs = reinterpret_cast<float*>(c.ws);
  if (p.splitk > 1) {
    size_t need = p.slab_bytes + (c.sync ? 0 : GEMM_SYNC_BYTES);
    DIHIP_REQUIRE(c.ws && c.ws_bytes >= need, DIHIP_MEMORY_ERROR,
                  "gemm_lowp: workspace too small (%zu < %zu)", c.ws_bytes, need);
    if (!c.sync) {
      counters = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(c.ws) + p.slab_bytes);
      DIHIP_CHECK_HIP(hipMemsetAsync(counters, 0, (size_t)p.col_blocks * p.m_blocks * sizeof(unsigned), stream),
                      DIHIP_RUNTIME_ERROR);
    }
  }
  GemmArgs a{};
  a.w0 = reinterpret_cast<const u32x4_t*>(c.w0);
  a.w1 = reinterpret_cast<const u32x4_t*>(c.w1);
  a.sz0 = reinterpret_cast<const uint32_t*>(c.sz0);
  a.sz1 = reinterpret_cast<const uint32_t*>(c.sz1);
  a.x = c.x;
  a.ldx = c.ldx;
  a.gamma = c.gamma;
  a.eps = c.eps;
  a.slabs = slabs;
  a.counters = counters;
  a.bias = c.bias;
  a.residual = c.residual;
  a.y = c.y;
  a.ldy = c.N;
  a.h_res = c.h_res;
  a.h_out = c.h_out;
  a.alpha = c.alpha;
  a.act = c.act;
  a.M = c.M;
  a.N = c.N;
  a.K = c.K;
  a.Np = d.Np;
  a.KT = d.KT;
  a.NTILES = d.NTILES;
  a.ksteps_per_group = d.group ? d.group / 32 : (1 << 30);
  a.Gp = lowp_dims(4, c.N, c.K, c.group_size).Gp;  // the pack kernel sizes Gp for the W4 padding
  a.splitk = p.splitk;
  a.ktiles_per_split = p.ktiles_per_split;
  a.kslice_tiles = p.kslice_tiles;
  const dim3 grid(p.col_blocks, p.splitk, p.m_blocks);
  hipError_t e = hipErrorInvalidValue;
  if (c.wbits == 4 && c.dtype == DIHIP_BF16) e = dispatch<4, DIHIP_BF16>(p, c.pro, c.epi, a, grid, stream);
  else if (c.wbits == 4 && c.dtype == DIHIP_F16) e = dispatch<4, DIHIP_F16>(p, c.pro, c.epi, a, grid, stream);
  else if (c.wbits == 8 && c.dtype == DIHIP_BF16) e = dispatch<8, DIHIP_BF16>(p, c.pro, c.epi, a, grid, stream);
  else if (c.wbits == 8 && c.dtype == DIHIP_F16) e = dispatch<8, DIHIP_F16>(p, c.pro, c.epi, a, grid, stream);
  else if (c.wbits == 16 && c.dtype == DIHIP_BF16) e = dispatch_dense<DIHIP_BF16>(p, c.pro, c.epi, a, grid, stream);
  else if (c.wbits == 16 && c.dtype == DIHIP_F16) e = dispatch_dense<DIHIP_F16>(p, c.pro, c.epi, a, grid, stream);
  DIHIP_REQUIRE(e == hipSuccess, e == hipErrorInvalidValue ? DIHIP_PARAM_ERROR : DIHIP_RUNTIME_ERROR,
                "gemm_lowp: launch failed (wbits=%d dtype=%d MT=%d NT=%d pro=%d epi=%d): %s", c.wbits, c.dtype,
                p.MT, p.NT, c.pro, c.epi, hipGetErrorString(e));
  return DIHIP_SUCCESS;
}

// true when a PRO_PLAIN bf16 call of this shape runs on the small-batch kernel (which reads FRAG32 faster)
static bool batch_kernel_eligible(int wbits, int M, int N, int K, int group_size) {
  if ((wbits != 4 && wbits != 8) || M <= 1 || M > 32 || N <= 0 || K <= 0) return false;
  if (!gemv_stream_enabled()) return false;
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  return K == d.Kp && (d.group == 0 || d.group % d.KTILE == 0);
}
static bool batch_kernel_shape(int wbits, int M, int N, int K, int group_size, bool dual) {
  return batch_kernel_eligible(wbits, M, N, K, group_size) && !make_gemv_plan(wbits, M, N, K, group_size, dual).ok;
}

// f32 hidden rows -> FT normalised rows (used by the fused entry points when M > 4); frag_mt > 0: FRAG32 output.
// One workgroup per row, the row stays in registers (one 16-byte load per 4 columns, all in flight together):
// a single round trip instead of two dependent passes over h.  cols % 4 == 0, cols <= 256 * 4 * VPT.
template <int FT, int VPT, int THREADS>
__global__ __launch_bounds__(THREADS) void rmsnorm_f32_to_ft_kernel(uint16_t* __restrict__ y, const float* __restrict__ h,
                                                                const void* __restrict__ gamma, float eps, int cols,
                                                                int frag_mt) {
  __shared__ float red[THREADS / 64];
  const int row = blockIdx.x, tid = threadIdx.x;
  const f32x4_t* hr = reinterpret_cast<const f32x4_t*>(h + (size_t)row * cols);
  const int nvec = cols >> 2;
  f32x4_t v[VPT];
  u32x2_t gm[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = tid + i * THREADS;
    v[i] = c < nvec ? hr[c] : f32x4_t{0.f, 0.f, 0.f, 0.f};
    gm[i] = c < nvec ? reinterpret_cast<const u32x2_t*>(gamma)[c] : u32x2_t{0u, 0u};
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) ss += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  float tot = red[0];
#pragma unroll
  for (int w = 1; w < THREADS / 64; ++w) tot += red[w];
  const float rstd = 1.f / sqrtf(tot / (float)cols + eps);
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = tid + i * THREADS;
    if (c < nvec) {
      const float g0 = ft_bits_to_f32<FT>(gm[i][0] & 0xFFFFu), g1 = ft_bits_to_f32<FT>(gm[i][0] >> 16);
      const float g2 = ft_bits_to_f32<FT>(gm[i][1] & 0xFFFFu), g3 = ft_bits_to_f32<FT>(gm[i][1] >> 16);
      const uint32_t lo = f32_to_ft_bits<FT>((g0 * v[i][0]) * rstd) | (f32_to_ft_bits<FT>((g1 * v[i][1]) * rstd) << 16);
      const uint32_t hi = f32_to_ft_bits<FT>((g2 * v[i][2]) * rstd) | (f32_to_ft_bits<FT>((g3 * v[i][3]) * rstd) << 16);
      const int k = c * 4;  // 4 consecutive columns stay adjacent in both layouts
      const size_t idx = frag_mt ? act_frag_index(row, k, frag_mt) : (size_t)row * cols + k;
      *reinterpret_cast<u32x2_t*>(y + idx) = u32x2_t{lo, hi};
    }
  }
}

// any width (two passes over the row)
template <int FT>
__global__ __launch_bounds__(256) void rmsnorm_f32_to_ft_wide_kernel(uint16_t* __restrict__ y, const float* __restrict__ h,
                                                                     const void* __restrict__ gamma, float eps, int cols,
                                                                     int frag_mt) {
  __shared__ float red[4];
  const float* hr = h + (size_t)blockIdx.x * cols;
  float ss = 0.f;
  for (int k = threadIdx.x; k < cols; k += 256) ss += hr[k] * hr[k];
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float rstd = 1.f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)cols + eps);
  for (int k = threadIdx.x; k < cols; k += 256) {
    const size_t idx = frag_mt ? act_frag_index((int)blockIdx.x, k, frag_mt) : (size_t)blockIdx.x * cols + k;
    y[idx] = (uint16_t)f32_to_ft_bits<FT>((load_ft<FT>(gamma, k) * hr[k]) * rstd);
  }
}

// row-major [M, K] 16-bit <-> FRAG32 (tests / callers that produce activations outside this library)
__global__ void act_frag_convert_kernel(uint16_t* __restrict__ dst, const uint16_t* __restrict__ src, int M, int K, int mt,
                                        int to_frag) {
  const size_t total = (size_t)M * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / K), k = (int)(i - (size_t)m * K);
    const size_t f = act_frag_index(m, k, mt);
    if (to_frag) dst[f] = src[i];
    else dst[i] = src[f];
  }
}

}  // namespace dihip

using namespace dihip;

extern "C" {

int dihip_debug_gemv_plan(int wbits, int M, int N, int K, int group_size, int dual, int* blocks, int* upb, int* wk, int* wn,
                          size_t* lds_bytes) {
  const GemvPlan p = make_gemv_plan(wbits, M, N, K, group_size, dual != 0);
  if (!p.ok) return DIHIP_PARAM_ERROR;
  if (blocks) *blocks = p.blocks;
  if (upb) *upb = p.upb;
  if (wk) *wk = p.WK;
  if (wn) *wn = p.WN;
  if (lds_bytes) *lds_bytes = p.lds_bytes;
  return DIHIP_SUCCESS;
}

size_t dihip_gemm_lowp_packed_weight_bytes(int wbits, int N, int K) {
  if ((wbits != 4 && wbits != 8) || N <= 0 || K <= 0) return 0;
  const LowpDims d = lowp_dims(wbits, N, K, -1);
  return (size_t)d.NTILES * d.KT * 64 * 16;
}

size_t dihip_gemm_lowp_packed_sz_bytes(int N, int K, int group_size) {
  if (N <= 0 || K <= 0) return 0;
  // Kp depends on wbits only through the tile (128 vs 64): size for the larger padding
  const LowpDims d = lowp_dims(4, N, K, group_size);
  return (size_t)d.Gp * d.Np * 4;
}

int dihip_gemm_lowp_pack(void* stream, int wbits, const void* wq, const void* scales, const void* zeros, int N,
                         int K, int group_size, int dtype, void* w_packed, void* sz_packed) {
  DIHIP_REQUIRE(wbits == 4 || wbits == 8, DIHIP_PARAM_ERROR, "gemm_lowp_pack: wbits must be 4 or 8");
  DIHIP_REQUIRE(N > 0 && K > 0 && wq && scales && zeros && w_packed && sz_packed, DIHIP_PARAM_ERROR,
                "gemm_lowp_pack: bad argument");
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "gemm_lowp_pack: dtype");
  DIHIP_REQUIRE(group_size <= 0 || group_size % 32 == 0, DIHIP_PARAM_ERROR,
                "gemm_lowp_pack: GroupSize %d must be a multiple of 32", group_size);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  const LowpDims d4 = lowp_dims(4, N, K, group_size);
  const size_t total = (size_t)d.NTILES * d.KT * 64 * 4;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 65535);
  if (wbits == 4)
    hipLaunchKernelGGL(pack_w4_kernel, dim3(blocks), dim3(256), 0, s, (const uint8_t*)wq, (uint32_t*)w_packed, N, K,
                       d.KT, d.NTILES);
  else
    hipLaunchKernelGGL(pack_w8_kernel, dim3(blocks), dim3(256), 0, s, (const int8_t*)wq, (uint32_t*)w_packed, N, K,
                       d.KT, d.NTILES);
  const size_t sztotal = (size_t)d4.Gp * d.Np;
  hipLaunchKernelGGL(pack_sz_kernel, dim3((int)std::min<size_t>((sztotal + 255) / 256, 65535)), dim3(256), 0, s,
                     (const uint16_t*)scales, (const uint16_t*)zeros, (uint32_t*)sz_packed, N, d.Np, d.G, d4.Gp);
  return launch_status();
}

size_t dihip_gemm_lowp_sync_bytes(void) { return GEMM_SYNC_BYTES; }

// K parts per tail tile the context-phase GEMM of this shape is launched with on this GPU (1: no tail split; diagnostics / tests)
int dihip_gemm_prefill_tail_parts(int wbits, int M, int N, int K, int group_size, int dual) {
  if (M < 64 || N <= 0 || K <= 0) return 1;
  return prefill_tail_plan(wbits, M, N, K, group_size, dual != 0).ksplit;
}

size_t dihip_gemm_lowp_workspace_bytes(int wbits, int M, int N, int K, int group_size) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  // sized for the dual (SwiGLU) form, the self-contained counter area and the M > 4 norm buffer
  const GemmPlan p1 = make_plan(wbits, M, N, K, group_size, false);
  const GemmPlan p2 = make_plan(wbits, M, N, K, group_size, true);
  size_t slab = std::max(p1.slab_bytes, p2.slab_bytes);
  if (wbits != 16 && M >= 64)
    slab = std::max(slab, std::max(prefill_tail_plan(wbits, M, N, K, group_size, false).slab_bytes, prefill_tail_plan(wbits, M, N, K, group_size, true).slab_bytes));
  if (wbits != 16 && M > 4 && M <= 32) {
    slab = std::max(slab, make_panel_plan(wbits, M, N, K, group_size, false).slab_bytes);
    slab = std::max(slab, make_panel_plan(wbits, M, N, K, group_size, true).slab_bytes);
    slab = std::max(slab, make_kslice_plan(wbits, M, N, K, group_size, true, true).slab_bytes);
  }
  return slab + GEMM_SYNC_BYTES + (size_t)((M + 15) / 16 * 16) * K * 2 + 512;  // norm rows may be FRAG32
}

static int gemm_std(void* stream, int wbits, const void* x, const void* w, const void* sz, const void* bias,
                    const void* residual, void* y, int M, int N, int K, int group_size, int act, float alpha,
                    void* ws, size_t ws_bytes, void* sync, int dtype) {
  GemmCall c{};
  c.wbits = wbits;
  c.dtype = dtype;
  c.pro = PRO_PLAIN;
  c.epi = EPI_STD;
  c.x = x;
  c.ldx = K;
  c.w0 = w;
  c.sz0 = sz;
  c.bias = bias;
  c.residual = residual;
  c.y = y;
  c.M = M;
  c.N = N;
  c.K = K;
  c.group_size = group_size;
  c.act = act;
  c.alpha = alpha;
  c.ws = ws;
  c.ws_bytes = ws_bytes;
  c.sync = sync;
  DIHIP_REQUIRE(y != nullptr || M == 0, DIHIP_PARAM_ERROR, "gemm_lowp: null output");
  return run_gemm(reinterpret_cast<hipStream_t>(stream), c);
}

int dihip_gemm_a16w8(void* stream, const void* x, const void* w_packed, const void* sz_packed, const void* bias,
                     const void* residual, void* y, int M, int N, int K, int group_size, int act, float alpha,
                     void* ws, size_t ws_bytes, void* sync, int dtype) {
  return gemm_std(stream, 8, x, w_packed, sz_packed, bias, residual, y, M, N, K, group_size, act, alpha, ws,
                  ws_bytes, sync, dtype);
}

int dihip_gemm_a16w4(void* stream, const void* x, const void* w_packed, const void* sz_packed, const void* bias,
                     const void* residual, void* y, int M, int N, int K, int group_size, int act, float alpha,
                     void* ws, size_t ws_bytes, void* sync, int dtype) {
  return gemm_std(stream, 4, x, w_packed, sz_packed, bias, residual, y, M, N, K, group_size, act, alpha, ws,
                  ws_bytes, sync, dtype);
}

// RMSNorm of M rows of the f32 hidden stream into FT rows (row-major or FRAG32)
static int launch_rmsnorm_rows(hipStream_t s, const float* h, const void* gamma, float eps, int M, int K, void* out, int frag_mt,
                               int dtype = DIHIP_BF16) {
  const bool vec = K % 4 == 0 && (reinterpret_cast<uintptr_t>(h) % 16 == 0) && (reinterpret_cast<uintptr_t>(gamma) % 8 == 0);
  uint16_t* xo = reinterpret_cast<uint16_t*>(out);
#define DIHIP_RMS_ROWS(FT_)                                                                                                        \
  do {                                                                                                                             \
    if (vec && K <= 4096)                                                                                                          \
      hipLaunchKernelGGL((rmsnorm_f32_to_ft_kernel<FT_, 4, 256>), dim3(M), dim3(256), 0, s, xo, h, gamma, eps, K, frag_mt);         \
    else if (vec && K <= 8192) /* wide rows: 1024 threads, two vectors each (256 threads took 6.8 us for 16 rows of 8192) */       \
      hipLaunchKernelGGL((rmsnorm_f32_to_ft_kernel<FT_, 2, 1024>), dim3(M), dim3(1024), 0, s, xo, h, gamma, eps, K, frag_mt);       \
    else                                                                                                                           \
      hipLaunchKernelGGL(rmsnorm_f32_to_ft_wide_kernel<FT_>, dim3(M), dim3(256), 0, s, xo, h, gamma, eps, K, frag_mt);              \
  } while (0)
  if (dtype == DIHIP_F16) DIHIP_RMS_ROWS(DIHIP_F16);
  else DIHIP_RMS_ROWS(DIHIP_BF16);
#undef DIHIP_RMS_ROWS
  return launch_status();
}

// For M > 4 the norm runs as its own small kernel into the tail of `ws`.
// The normalised rows are private to the call, so they are written in whatever layout the GEMM kernel that
// follows reads fastest (*x_layout).
static int norm_to_ws(hipStream_t s, const float* h, const void* gamma, float eps, int M, int K, int dtype, void* ws,
                      size_t ws_bytes, int wbits, int N, int group_size, bool dual, void** xnorm, size_t* ws_left,
                      int* x_layout, bool force_frag = false) {
  const GemmPlan p = make_plan(wbits, M, N, K, group_size, dual);
  size_t slab = p.slab_bytes;
  if (wbits != 16 && M >= 64) slab = std::max(slab, prefill_tail_plan(wbits, M, N, K, group_size, dual).slab_bytes);
  if (wbits != 16 && M > 4 && M <= 32) {
    slab = std::max(slab, make_panel_plan(wbits, M, N, K, group_size, dual).slab_bytes);
    slab = std::max(slab, make_kslice_plan(wbits, M, N, K, group_size, dual, true).slab_bytes);
  }
  const size_t off = (slab + 255) & ~(size_t)255;
  const bool frag = force_frag || batch_kernel_shape(wbits, M, N, K, group_size, dual);  // (bf16 and f16: both instantiated since round 5)
  const int mt = M > 16 ? 2 : 1;
  const size_t xbytes = frag ? (size_t)mt * 16 * K * 2 : (size_t)M * K * 2;
  DIHIP_REQUIRE(ws && ws_bytes >= off + xbytes, DIHIP_MEMORY_ERROR, "fused gemm: workspace too small");
  *xnorm = reinterpret_cast<char*>(ws) + off;
  *ws_left = off;
  *x_layout = frag ? DIHIP_ACT_FRAG32 : DIHIP_ACT_ROWMAJOR;
  return launch_rmsnorm_rows(s, h, gamma, eps, M, K, *xnorm, frag ? mt : 0, dtype);
}

int dihip_fused_norm_gemm(void* stream, int wbits, const float* h, const void* gamma, float eps, const void* w_packed,
                          const void* sz_packed, const void* bias, void* y, int M, int N, int K, int group_size,
                          int act, void* ws, size_t ws_bytes, void* sync, int dtype) {
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "fused path: bf16 or f16 activations");
  DIHIP_REQUIRE(sync != nullptr, DIHIP_PARAM_ERROR, "fused path needs a sync buffer");
  if (M == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  GemmCall c{};
  c.wbits = wbits;
  c.dtype = dtype;
  c.epi = EPI_STD;
  c.w0 = w_packed;
  c.sz0 = sz_packed;
  c.bias = bias;
  c.y = y;
  c.M = M;
  c.N = N;
  c.K = K;
  c.group_size = group_size;
  c.act = act;
  c.alpha = 1.f;
  c.sync = sync;
  c.ldx = K;
  if (M <= 4) {
    c.pro = PRO_RMSNORM;
    c.x = h;
    c.gamma = gamma;
    c.eps = eps;
    c.ws = ws;
    c.ws_bytes = ws_bytes;
  } else {
    void* xn;
    size_t left;
    int st = norm_to_ws(s, h, gamma, eps, M, K, dtype, ws, ws_bytes, wbits, N, group_size, false, &xn, &left, &c.x_layout);
    if (st) return st;
    c.pro = PRO_PLAIN;
    c.x = xn;
    c.ws = ws;
    c.ws_bytes = left;
  }
  return run_gemm(s, c);
}

int dihip_fused_norm_swiglu(void* stream, int wbits, const float* h, const void* gamma, float eps,
                            const void* wg_packed, const void* szg_packed, const void* wu_packed,
                            const void* szu_packed, void* y, int M, int N, int K, int group_size, void* ws,
                            size_t ws_bytes, void* sync, int dtype) {
  return dihip_fused_norm_swiglu_ex(stream, wbits, h, gamma, eps, wg_packed, szg_packed, wu_packed, szu_packed, y, M, N, K,
                                    group_size, ws, ws_bytes, sync, dtype, DIHIP_ACT_ROWMAJOR);
}

int dihip_fused_norm_swiglu_ex(void* stream, int wbits, const float* h, const void* gamma, float eps,
                               const void* wg_packed, const void* szg_packed, const void* wu_packed,
                               const void* szu_packed, void* y, int M, int N, int K, int group_size, void* ws,
                               size_t ws_bytes, void* sync, int dtype, int y_layout) {
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "fused path: bf16 or f16 activations");
  DIHIP_REQUIRE(y_layout == DIHIP_ACT_ROWMAJOR || y_layout == DIHIP_ACT_FRAG32, DIHIP_PARAM_ERROR, "fused swiglu: bad y_layout");
  DIHIP_REQUIRE(y_layout == DIHIP_ACT_ROWMAJOR || (M > 4 && batch_kernel_eligible(wbits, M, N, K, group_size)), DIHIP_PARAM_ERROR,
                "fused swiglu: FRAG32 output needs the small-batch kernel (4 < M <= 32, K a multiple of the k-tile)");
  DIHIP_REQUIRE(sync != nullptr, DIHIP_PARAM_ERROR, "fused path needs a sync buffer");
  if (M == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  GemmCall c{};
  c.wbits = wbits;
  c.dtype = dtype;
  c.epi = EPI_SWIGLU;
  c.w0 = wg_packed;
  c.sz0 = szg_packed;
  c.w1 = wu_packed;
  c.sz1 = szu_packed;
  c.y = y;
  c.M = M;
  c.N = N;
  c.K = K;
  c.group_size = group_size;
  c.alpha = 1.f;
  c.sync = sync;
  c.ldx = K;
  c.y_layout = y_layout;
  if (M <= 4) {
    c.pro = PRO_RMSNORM;
    c.x = h;
    c.gamma = gamma;
    c.eps = eps;
    c.ws = ws;
    c.ws_bytes = ws_bytes;
  } else {
    void* xn;
    size_t left;
    int st = norm_to_ws(s, h, gamma, eps, M, K, dtype, ws, ws_bytes, wbits, N, group_size, true, &xn, &left, &c.x_layout,
                        y_layout == DIHIP_ACT_FRAG32);
    if (st) return st;
    c.pro = PRO_PLAIN;
    c.x = xn;
    c.ws = ws;
    c.ws_bytes = left;
  }
  return run_gemm(s, c);
}

int dihip_fused_gemm_addto(void* stream, int wbits, const void* x, const void* w_packed, const void* sz_packed,
                           const float* h_res, float* h_out, int M, int N, int K, int group_size, void* ws,
                           size_t ws_bytes, void* sync, int dtype) {
  return dihip_fused_gemm_addto_ex(stream, wbits, x, w_packed, sz_packed, h_res, h_out, M, N, K, group_size, ws, ws_bytes,
                                   sync, dtype, DIHIP_ACT_ROWMAJOR);
}

int dihip_gemm_lowp_prefers_frag(int wbits, int M, int N, int K, int group_size, int dual) {
  return batch_kernel_shape(wbits, M, N, K, group_size, dual != 0) ? 1 : 0;
}

size_t dihip_act_frag_bytes(int M, int K) {
  if (M <= 0 || M > 32 || K <= 0 || K % 32) return 0;
  return (size_t)(M > 16 ? 32 : 16) * K * 2;
}

static int act_convert(void* stream, const void* src, void* dst, int M, int K, int dtype, int to_frag) {
  DIHIP_REQUIRE(src && dst && M > 0 && M <= 32 && K > 0 && K % 32 == 0, DIHIP_PARAM_ERROR,
                "act_frag: need 0 < M <= 32 and K a multiple of 32");
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "act_frag: 16-bit activations only");
  const size_t total = (size_t)M * K;
  hipLaunchKernelGGL(act_frag_convert_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (uint16_t*)dst, (const uint16_t*)src, M, K, M > 16 ? 2 : 1, to_frag);
  return launch_status();
}
int dihip_act_to_frag(void* stream, const void* x_rowmajor, void* x_frag, int M, int K, int dtype) {
  return act_convert(stream, x_rowmajor, x_frag, M, K, dtype, 1);
}
int dihip_act_from_frag(void* stream, const void* x_frag, void* x_rowmajor, int M, int K, int dtype) {
  return act_convert(stream, x_frag, x_rowmajor, M, K, dtype, 0);
}

int dihip_fused_gemm_addto_ex(void* stream, int wbits, const void* x, const void* w_packed, const void* sz_packed,
                              const float* h_res, float* h_out, int M, int N, int K, int group_size, void* ws,
                              size_t ws_bytes, void* sync, int dtype, int x_layout) {
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "fused path: bf16 or f16 activations");
  DIHIP_REQUIRE(x_layout == DIHIP_ACT_ROWMAJOR || x_layout == DIHIP_ACT_FRAG32, DIHIP_PARAM_ERROR, "fused addto: bad x_layout");
  // h_res may be NULL: then h_out = x . W (row-parallel TP ranks other than 0, gemm_op.cpp:133-137)
  DIHIP_REQUIRE(sync != nullptr && h_out, DIHIP_PARAM_ERROR, "fused addto: null pointer");
  GemmCall c{};
  c.wbits = wbits;
  c.dtype = dtype;
  c.pro = PRO_PLAIN;
  c.epi = EPI_ADDTO;
  c.x = x;
  c.ldx = K;
  c.x_layout = x_layout;
  c.w0 = w_packed;
  c.sz0 = sz_packed;
  c.h_res = h_res;
  c.h_out = h_out;
  c.M = M;
  c.N = N;
  c.K = K;
  c.group_size = group_size;
  c.alpha = 1.f;
  c.ws = ws;
  c.ws_bytes = ws_bytes;
  c.sync = sync;
  return run_gemm(reinterpret_cast<hipStream_t>(stream), c);
}

static int gemm_addto_norm_impl(void* stream, int wbits, const void* x, const void* w_packed, const void* sz_packed,
                                const float* h_res, float* h_out, int M, int N, int K, int group_size, void* ws,
                                size_t ws_bytes, void* sync, int dtype, int x_layout, const void* gamma, float eps,
                                void* xnorm, int xnorm_layout, float* rowsq, size_t rowsq_bytes, int* rowsq_parts) {
  if (rowsq_parts) *rowsq_parts = 0;
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "fused path: bf16 or f16 activations");
  DIHIP_REQUIRE(x_layout == DIHIP_ACT_ROWMAJOR || x_layout == DIHIP_ACT_FRAG32, DIHIP_PARAM_ERROR, "fused addto: bad x_layout");
  DIHIP_REQUIRE(xnorm_layout == DIHIP_ACT_ROWMAJOR || (xnorm_layout == DIHIP_ACT_FRAG32 && M <= 32 && N % 32 == 0),
                DIHIP_PARAM_ERROR, "fused addto + norm: FRAG32 output needs M <= 32 and N %% 32 == 0");
  DIHIP_REQUIRE(sync != nullptr && h_out && gamma && xnorm, DIHIP_PARAM_ERROR, "fused addto + norm: null pointer");
  if (M == 0) return DIHIP_SUCCESS;
  bool done = false;
  GemmCall c{};
  c.wbits = wbits;
  c.dtype = dtype;
  c.pro = PRO_PLAIN;
  c.epi = EPI_ADDTO;
  c.x = x;
  c.ldx = K;
  c.x_layout = x_layout;
  c.w0 = w_packed;
  c.sz0 = sz_packed;
  c.h_res = h_res;
  c.h_out = h_out;
  c.M = M;
  c.N = N;
  c.K = K;
  c.group_size = group_size;
  c.alpha = 1.f;
  c.ws = ws;
  c.ws_bytes = ws_bytes;
  c.sync = sync;
  c.n_gamma = gamma;
  c.n_eps = eps;
  c.n_out = xnorm;
  c.n_frag_mt = xnorm_layout == DIHIP_ACT_FRAG32 ? (M > 16 ? 2 : 1) : 0;
  c.n_done = &done;
  c.n_rowsq = rowsq;
  c.n_rowsq_bytes = rowsq_bytes;
  c.n_parts = rowsq ? rowsq_parts : nullptr;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int st = run_gemm(s, c);
  if (st || done) return st;
  return launch_rmsnorm_rows(s, h_out, gamma, eps, M, N, xnorm, c.n_frag_mt, dtype);  // this plan has no slab reduction to ride on
}

int dihip_fused_gemm_addto_norm(void* stream, int wbits, const void* x, const void* w_packed, const void* sz_packed,
                                const float* h_res, float* h_out, int M, int N, int K, int group_size, void* ws,
                                size_t ws_bytes, void* sync, int dtype, int x_layout, const void* gamma, float eps,
                                void* xnorm, int xnorm_layout) {
  return gemm_addto_norm_impl(stream, wbits, x, w_packed, sz_packed, h_res, h_out, M, N, K, group_size, ws, ws_bytes, sync, dtype, x_layout,
                              gamma, eps, xnorm, xnorm_layout, nullptr, 0, nullptr);
}

// ... with the RMSNorm DEFERRED where the serving kernel offers it (gemv_batch_kernel.hpp): *rowsq_parts > 0 -- xnorm holds
// FT(gamma * h_out) and rowsq[part][32] the partial sums of h_out^2, to be handed to dihip_prenorm_{gemm,swiglu}_rowsq;
// *rowsq_parts == 0 -- xnorm is the finished norm (as dihip_fused_gemm_addto_norm)
int dihip_fused_gemm_addto_prenorm(void* stream, int wbits, const void* x, const void* w_packed, const void* sz_packed,
                                   const float* h_res, float* h_out, int M, int N, int K, int group_size, void* ws,
                                   size_t ws_bytes, void* sync, int dtype, int x_layout, const void* gamma, float eps,
                                   void* xnorm, int xnorm_layout, float* rowsq, size_t rowsq_bytes, int* rowsq_parts) {
  DIHIP_REQUIRE(rowsq && rowsq_parts && reinterpret_cast<uintptr_t>(rowsq) % 16 == 0, DIHIP_PARAM_ERROR, "fused addto + prenorm: null / misaligned rowsq");
  return gemm_addto_norm_impl(stream, wbits, x, w_packed, sz_packed, h_res, h_out, M, N, K, group_size, ws, ws_bytes, sync, dtype, x_layout,
                              gamma, eps, xnorm, xnorm_layout, rowsq, rowsq_bytes, rowsq_parts);
}

size_t dihip_rowsq_bytes() { return (size_t)256 * 32 * sizeof(float); }  // one part per workgroup of a producer (<= one per CU)

// can a prenorm GEMM (dual = 0) / SwiGLU pair (dual = 1) of this shape take deferred row norms?  (mirrors run_gemm's dispatch)
int dihip_prenorm_rowsq_supported(int wbits, int M, int N, int K, int group_size, int dual, int dtype, int x_layout) {
  static const bool on = !env_off("DIHIP_DEFER_RMSNORM");  // =0: never (A/B; callers then keep the norm launch)
  if (!on || M <= 4 || M > 32 || dtype != DIHIP_BF16 || (wbits != 4 && wbits != 8) || !gemv_stream_enabled()) return 0;
  const LowpDims d = lowp_dims(wbits, N, K, group_size);
  if (K != d.Kp || !(d.group == 0 || d.group % d.KTILE == 0)) return 0;
  if (x_layout == DIHIP_ACT_FRAG32) {
    const KslicePlan kp = make_kslice_plan(wbits, M, N, K, group_size, dual != 0);
    if (kp.ok) return 1;
    const PanelPlan pp = make_panel_plan(wbits, M, N, K, group_size, dual != 0);
    if (pp.ok) return pp.nslices > 1 ? 1 : 0;
  }
  return dual ? 0 : 1;  // gemv_batch_kernel: plain epilogue only
}

// LayerNormNoBeta of the f32 hidden rows into FT rows, on its own (the MoE layer feeds four consumers from it)
int dihip_rmsnorm_rows(void* stream, void* xnorm, const float* h, const void* gamma, float eps, int M, int K, int dtype) {
  DIHIP_REQUIRE(M >= 0 && K > 0 && xnorm && h && gamma, DIHIP_PARAM_ERROR, "rmsnorm_rows: bad argument");
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "rmsnorm_rows: 16-bit rows only");
  if (M == 0) return DIHIP_SUCCESS;
  return launch_rmsnorm_rows(reinterpret_cast<hipStream_t>(stream), h, gamma, eps, M, K, xnorm, 0, dtype);
}

int dihip_prenorm_gemm(void* stream, int wbits, const void* xnorm, int x_layout, const void* w_packed, const void* sz_packed,
                       const void* bias, void* y, int M, int N, int K, int group_size, int act, void* ws, size_t ws_bytes,
                       void* sync, int dtype) {
  return dihip_prenorm_gemm_rowsq(stream, wbits, xnorm, x_layout, w_packed, sz_packed, bias, y, M, N, K, group_size, act, ws, ws_bytes, sync,
                                  dtype, nullptr, 0, 0.f);
}

int dihip_prenorm_gemm_rowsq(void* stream, int wbits, const void* xnorm, int x_layout, const void* w_packed, const void* sz_packed,
                             const void* bias, void* y, int M, int N, int K, int group_size, int act, void* ws, size_t ws_bytes,
                             void* sync, int dtype, const float* rowsq, int rowsq_parts, float eps) {
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "fused path: bf16 or f16 activations");
  DIHIP_REQUIRE(x_layout == DIHIP_ACT_ROWMAJOR || x_layout == DIHIP_ACT_FRAG32, DIHIP_PARAM_ERROR, "prenorm gemm: bad x_layout");
  DIHIP_REQUIRE(sync != nullptr && xnorm && y, DIHIP_PARAM_ERROR, "prenorm gemm: null pointer");
  GemmCall c{};
  c.wbits = wbits;
  c.dtype = dtype;
  c.pro = PRO_PLAIN;
  c.epi = EPI_STD;
  c.x = xnorm;
  c.ldx = K;
  c.x_layout = x_layout;
  c.w0 = w_packed;
  c.sz0 = sz_packed;
  c.bias = bias;
  c.y = y;
  c.M = M;
  c.N = N;
  c.K = K;
  c.group_size = group_size;
  c.act = act;
  c.alpha = 1.f;
  c.ws = ws;
  c.ws_bytes = ws_bytes;
  c.sync = sync;
  c.rowsq = rowsq_parts > 0 ? rowsq : nullptr;
  c.rowsq_parts = rowsq_parts;
  c.rowsq_eps = eps;
  return run_gemm(reinterpret_cast<hipStream_t>(stream), c);
}

int dihip_prenorm_swiglu(void* stream, int wbits, const void* xnorm, int x_layout, const void* wg_packed,
                         const void* szg_packed, const void* wu_packed, const void* szu_packed, void* y, int M, int N, int K,
                         int group_size, void* ws, size_t ws_bytes, void* sync, int dtype, int y_layout) {
  return dihip_prenorm_swiglu_rowsq(stream, wbits, xnorm, x_layout, wg_packed, szg_packed, wu_packed, szu_packed, y, M, N, K, group_size, ws,
                                    ws_bytes, sync, dtype, y_layout, nullptr, 0, 0.f);
}

int dihip_prenorm_swiglu_rowsq(void* stream, int wbits, const void* xnorm, int x_layout, const void* wg_packed,
                               const void* szg_packed, const void* wu_packed, const void* szu_packed, void* y, int M, int N, int K,
                               int group_size, void* ws, size_t ws_bytes, void* sync, int dtype, int y_layout, const float* rowsq,
                               int rowsq_parts, float eps) {
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "fused path: bf16 or f16 activations");
  DIHIP_REQUIRE(x_layout == DIHIP_ACT_ROWMAJOR || x_layout == DIHIP_ACT_FRAG32, DIHIP_PARAM_ERROR, "prenorm swiglu: bad x_layout");
  DIHIP_REQUIRE(y_layout == DIHIP_ACT_ROWMAJOR || y_layout == DIHIP_ACT_FRAG32, DIHIP_PARAM_ERROR, "prenorm swiglu: bad y_layout");
  DIHIP_REQUIRE(sync != nullptr && xnorm && y, DIHIP_PARAM_ERROR, "prenorm swiglu: null pointer");
  DIHIP_REQUIRE(y_layout == DIHIP_ACT_ROWMAJOR || (M > 4 && batch_kernel_eligible(wbits, M, N, K, group_size)), DIHIP_PARAM_ERROR,
                "prenorm swiglu: FRAG32 output needs the small-batch kernel (4 < M <= 32, K a multiple of the k-tile)");
  GemmCall c{};
  c.wbits = wbits;
  c.dtype = dtype;
  c.pro = PRO_PLAIN;
  c.epi = EPI_SWIGLU;
  c.x = xnorm;
  c.ldx = K;
  c.x_layout = x_layout;
  c.y_layout = y_layout;
  c.w0 = wg_packed;
  c.sz0 = szg_packed;
  c.w1 = wu_packed;
  c.sz1 = szu_packed;
  c.y = y;
  c.M = M;
  c.N = N;
  c.K = K;
  c.group_size = group_size;
  c.alpha = 1.f;
  c.ws = ws;
  c.ws_bytes = ws_bytes;
  c.sync = sync;
  c.rowsq = rowsq_parts > 0 ? rowsq : nullptr;
  c.rowsq_parts = rowsq_parts;
  c.rowsq_eps = eps;
  return run_gemm(reinterpret_cast<hipStream_t>(stream), c);
}


// ---- unquantised 16-bit weights (lm_head / dense Gemm), same kernel family ----------------------
size_t dihip_dense_packed_weight_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  const LowpDims d = lowp_dims(16, N, K, -1);
  return (size_t)d.NTILES * d.KT * 64 * 16;
}

int dihip_dense_pack(void* stream, const void* w_kn, int N, int K, int dtype, void* w_packed) {
  DIHIP_REQUIRE(N > 0 && K > 0 && w_kn && w_packed, DIHIP_PARAM_ERROR, "dense_pack: bad argument");
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "dense_pack: dtype");
  const LowpDims d = lowp_dims(16, N, K, -1);
  const size_t total = (size_t)d.NTILES * d.KT * 64 * 4;
  hipLaunchKernelGGL(pack_w16_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (const uint16_t*)w_kn, (uint32_t*)w_packed, N, K, d.KT,
                     d.NTILES);
  return launch_status();
}

size_t dihip_dense_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const GemmPlan p = make_plan(16, M, N, K, -1, false);
  return p.slab_bytes + GEMM_SYNC_BYTES + (size_t)M * K * 2 + 512;
}

int dihip_gemm_a16w16(void* stream, const void* x, const void* w_packed, const void* bias, const void* residual,
                      void* y, int M, int N, int K, int act, float alpha, void* ws, size_t ws_bytes, void* sync,
                      int dtype) {
  GemmCall c{};
  c.wbits = 16;
  c.dtype = dtype;
  c.pro = PRO_PLAIN;
  c.epi = EPI_STD;
  c.x = x;
  c.ldx = K;
  c.w0 = w_packed;
  c.bias = bias;
  c.residual = residual;
  c.y = y;
  c.M = M;
  c.N = N;
  c.K = K;
  c.group_size = -1;
  c.act = act;
  c.alpha = alpha;
  c.ws = ws;
  c.ws_bytes = ws_bytes;
  c.sync = sync;
  return run_gemm(reinterpret_cast<hipStream_t>(stream), c);
}

int dihip_lm_head(void* stream, float* logits, const float* h, const void* gamma, float eps, const void* w_packed,
                  int M, int N, int K, void* ws, size_t ws_bytes, void* sync, int dtype) {
  DIHIP_REQUIRE(dtype == DIHIP_BF16 || dtype == DIHIP_F16, DIHIP_PARAM_ERROR, "lm_head: dtype");
  DIHIP_REQUIRE(logits && h && w_packed && sync, DIHIP_PARAM_ERROR, "lm_head: null pointer");
  if (M == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  GemmCall c{};
  c.wbits = 16;
  c.dtype = dtype;
  c.epi = EPI_ADDTO;
  c.w0 = w_packed;
  c.h_res = nullptr;
  c.h_out = logits;
  c.M = M;
  c.N = N;
  c.K = K;
  c.group_size = -1;
  c.alpha = 1.f;
  c.sync = sync;
  c.ldx = K;
  if (gamma != nullptr && M <= 4) {
    c.pro = PRO_RMSNORM;
    c.x = h;
    c.gamma = gamma;
    c.eps = eps;
    c.ws = ws;
    c.ws_bytes = ws_bytes;
  } else {
    DIHIP_REQUIRE(gamma != nullptr, DIHIP_PARAM_ERROR, "lm_head: needs gamma");
    void* xn;
    size_t left;
    int st = norm_to_ws(s, h, gamma, eps, M, K, dtype, ws, ws_bytes, 16, N, -1, false, &xn, &left, &c.x_layout);
    if (st) return st;
    c.pro = PRO_PLAIN;
    c.x = xn;
    c.ws = ws;
    c.ws_bytes = left;
  }
  return run_gemm(s, c);
}

}  // extern "C"
