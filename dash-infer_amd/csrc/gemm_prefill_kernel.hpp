// gemm_prefill_kernel.hpp -- weight-only (A16W8 / A16W4) GEMM for the CONTEXT phase (M >= 64 rows) on gfx950.
//
// The general kernel (gemm_lowp_kernel.hpp) keeps at most 32 activation rows per workgroup: a 2048-token prompt walks every
// weight matrix 64 times and the context phase of Qwen2-7B ran at 9 % of the dense bf16 MFMA peak (profiles/r03f_bench_prefill_2048.json:
// 119 ms, of which the attention is 1.5).  This kernel is the large-M member of the family -- replaces, for prefill shapes,
// the reference's dequantise + cuBLAS fall-back and its 16816 tensor-core kernels (gemm_a16w8_kernel.h:239-331,
// hgemm_a16w4_subc_32x256x32, gemm_a16w4_subc_kernel.cu:466-815):
//
//   * workgroup tile 128 rows x 256 columns (8 waves side by side in N, each 128 x 32 = 8 row tiles x 2 column tiles of
//     v_mfma_f32_16x16x32: 16 accumulators), K walked in packed k-tiles (W4: 128 k, W8: 64 k);
//   * A (activations, FT row-major) goes global -> registers -> LDS once per workgroup and k-tile, stored as the MFMA A
//     fragments (one fragment = one contiguous 1 KiB: conflict-free ds_write_b128 / ds_read_b128), double-buffered: the loads of
//     k-tile t + 1 are in flight while t is multiplied, ONE barrier per k-tile; every wave reads all 8 row tiles;
//   * B (weights) never touches LDS: "dihip tile-major" packing makes a lane's 16-byte load the B fragment of its own column
//     tile (x KSTEPS), and no two waves share a column tile;
//   * the same exact-integer arithmetic as the decode kernels: q expands to 128 + q in FT with the magic-number trick, the MFMA
//     sums exact products, and per quantisation group  y += s * (acc - (z + 128) * sum_k x)  on the f32 accumulator, with
//     sum_k x per (row, k-tile) taken while A is staged -- four MFMAs against a fragment of ones per wave and k-tile (the VALU
//     form cost 72 instructions per wave and k-tile: +3 % tokens/s);
//   * epilogues as everywhere: STD (alpha, bias, UnaryType activation, FT residual), SwiGLU over a gate / up pair (a wave's two
//     column tiles are then the SAME 16 columns of the two matrices: 128 output columns per workgroup), f32 hidden-stream update.
//
// Host contract (gemm_lowp.hip: run_gemm): bf16 / f16 activations row-major with ldx % 8 == 0, 16-byte aligned; K a multiple of
// the k-tile; group_size a multiple of the k-tile or per-channel.  Rows >= M are clamped on load and masked on store.
//
// Where the time goes (round 3, SQ counters of the SwiGLU launch, profiles/r03_prefill_gemm_sq_counters.txt): 244 registers mean
// one workgroup = two waves per SIMD; per wave and k-tile 68 MFMAs (16 cycles of the matrix pipe each) + 193 other VALU + 45 LDS +
// 32 scalar instructions.  MFMA and VALU share the SIMD's one VALU-class issue slot per 4 cycles, so the two waves need
// 2 x 261 x 4 = 2088 issue cycles next to 2176 cycles of matrix pipe: the loop only runs at the MFMA rate if every gap of every
// MFMA carries exactly its three VALU instructions.  Measured: matrix pipe busy 39 % of the kernel (790-870 TFLOP/s), waves
// issuing 31 %, issue-stalled 44 %, parked at waitcnt / barrier 25 %.  Tried and measured, none better than +-3 %: the loop
// without per-k-step scheduling fences (kept, +3 % with s_setprio around the multiply), an explicit sched_group_barrier
// interleave with software-pipelined expansion, a 2 x 4 wave grid (half the LDS reads, twice the expansions: -10 %), a ping-pong
// schedule (two groups of four waves half a k-tile period apart so that each SIMD always has one multiplying wave: -7 %, or
// -2 % with scalar instead of SLP-packed FMAs), scalar FMAs alone (-2 %).  The instruction mix is the limit: the fix-up
// (2 FMA per accumulator element and group) and the 4-bit expansion (7 VALU per 8 weights) are inherent to multiplying the
// integer weights in place.
// Where the rest goes (per-wave cycle stamps + tools/gemm_skeleton_bench, same profile): the multiply phase itself runs at the
// rate of the bare skeleton -- LDS fragment reads + MFMAs + barrier with nothing else reach 70 % of the matrix peak in this
// tile shape, 82 % with the reads of k-step s + 1 issued ahead of the MFMAs of s (2 657 / 2 454 cycles per k-tile; the kernel's
// two waves multiply for 2 620-3 312) -- and then ~1 600 cycles per k-tile pass with the pipe idle: fix-up 564, staging 408,
// barrier 628.  Next: stage A with direct global -> LDS loads (no staging phase, 16 registers back; the row-sum MFMAs move to
// the multiply, which reads those fragments anyway), spend the registers on pipelined fragment reads, and slide the fix-up under
// the first k-step of the next tile.  (A one-wave-per-SIMD 128 x 128 wave tile measured WORSE as a bare skeleton: 54 %.)
#pragma once
#include <atomic>
#include "gemm_lowp_kernel.hpp"
#include "gemv_stream_kernel.hpp"  // ExpandV: the 128 + q expansion of the decode kernels

namespace dihip {

// what-if timing builds (tools/build_ksl_variant.sh NAME -DDIHIP_PF_X=.. gemm_prefill_inst; results WRONG with any bit set):
//   1 no scale / zero-point fix-up   2 no A staging (LDS keeps what the prologue wrote)   4 no barrier   8 no nibble expansion
// Round 4, the SwiGLU GEMM of a 2048-token prompt (profiles/r04af_*): 676 us as built (822 TFLOP/s); 645 without (1), 597 without (2),
// 649 without (4), 654 without (8); 459 without all four (1 212 TFLOP/s = 48 % of the dense peak): the bare LDS-read + MFMA loop
// of this tile shape is the ceiling, and the A staging (registers -> ds_write + the row-sum MFMAs) is the largest single extra.
#ifndef DIHIP_PF_X
#define DIHIP_PF_X 0
#endif
constexpr int PF_WAVES = 8;
constexpr int PF_THREADS = PF_WAVES * 64;
constexpr int PF_BM = 128;           // rows per workgroup (8 row tiles)
constexpr int PF_WM = 1;             // waves along M (a 2 x 4 grid with 4 x 4 tiles per wave measured 10 % slower: twice the expansions)
constexpr int PF_WN = PF_WAVES / PF_WM;
constexpr int PF_TR = PF_BM / 16;    // row tiles of the workgroup (= waves: wave w stages row tile w)
constexpr int PF_RT = PF_TR / PF_WM; // row tiles per wave
constexpr int PF_CW = 2 * PF_WM;     // column tiles per wave (the workgroup covers 256 columns either way)
static_assert(PF_TR == PF_WAVES, "one staged row tile per wave");

struct PrefillArgs {
  const u32x4_t* w0;
  const u32x4_t* w1;    // EPI_SWIGLU: "up" weight
  const uint32_t* sz0;  // [NTILES][Gp][16] (scale | zero << 16)
  const uint32_t* sz1;
  const void* x;        // FT [M, ldx]
  int ldx;
  const void* bias;
  const void* residual;  // FT [M, ldy]
  void* y;               // FT [M, ldy]
  int ldy;
  const float* h_res;  // EPI_ADDTO
  float* h_out;
  float alpha;
  int act;
  int M, N;
  int KT;      // k-tiles
  int NTILES;  // 16-column tiles
  int Gp;      // (scale, zero) groups stored per tile
  int ktpg;    // k-tiles per quantisation group (per-channel: >= KT)
  int col_blocks;
  // Tail split (round 5): column blocks [tail_cb, col_blocks) -- the workgroups that would otherwise run as a last, mostly empty
  // round on the chip (qkv of Qwen2-7B at 2048 rows: 288 tiles on 256 CUs) -- are walked by `ksplit` workgroups each, one
  // K range apiece (whole quantisation groups), which leave their f32 tile in `slab`; gemm_prefill_tail_reduce_kernel adds the
  // parts in order and applies the epilogue.  tail_cb == col_blocks: no split.
  int tail_cb;
  int ksplit;
  float* slab;  // [tail tile][ksplit][PF_BM][256] f32
  unsigned long long* trace;  // per-wave cycle stamps (library built with -DDIHIP_PF_TRACE, tools/prefill_trace.py)
};

template <int WBITS>
constexpr size_t prefill_lds_bytes() {
  // 2 x A tile (PF_TR x KSTEPS fragments of 1 KiB) + 2 x 128 row sums
  return (size_t)2 * PF_TR * WTraits<WBITS>::KSTEPS * 1024 + 2 * PF_BM * sizeof(float);
}

// (GPT -- one k-tile per quantisation group -- is a compile-time copy of the same code: the group test folds away)
template <int WBITS, int FT, int EPI, int GPT>
__global__ __launch_bounds__(PF_THREADS, 2) void gemm_prefill_kernel(const PrefillArgs a) {
  using WT = WTraits<WBITS>;
  using EX = ExpandV<WBITS, FT>;
  constexpr int KSTEPS = WT::KSTEPS;
  constexpr int KTILE = WT::KTILE;
  constexpr bool DUAL = EPI == EPI_SWIGLU;
  constexpr size_t ABYTES = (size_t)PF_TR * KSTEPS * 1024;  // one A tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* xsum = reinterpret_cast<float*>(smem + 2 * ABYTES);  // [2][PF_BM]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ni = lane & 15, kb = lane >> 4;
  const int wm = wave / PF_WN, wn = wave - wm * PF_WN;  // this wave's place in the PF_WM x PF_WN grid
  // consecutive workgroups share a column block (its weights stay hot in the L2s) and walk the rows
  const int mblocks = (a.M + PF_BM - 1) / PF_BM;
  int cb = blockIdx.x / mblocks, mb = blockIdx.x - cb * mblocks;
  int kt0 = 0, kt1 = a.KT, split_part = -1, split_tile = 0;  // this workgroup's k-tiles; tail split: its part and tile
  if (cb >= a.tail_cb) {  // (uniform) mb fastest, then the K part, then the column block: the 16 row blocks of a part share its weights
    const int idx = (int)blockIdx.x - a.tail_cb * mblocks;
    mb = idx % mblocks;
    const int t = idx / mblocks;
    split_part = t % a.ksplit;
    const int cbt = t / a.ksplit;
    cb = a.tail_cb + cbt;
    const int per = a.KT / a.ksplit;
    kt0 = split_part * per;
    kt1 = kt0 + per;
    split_tile = cbt * mblocks + mb;
  }
  const int m0 = mb * PF_BM;

  // ---- this wave's column tiles -------------------------------------------------------------------
  // STD / ADDTO: PF_CW consecutive tiles of the one matrix;  SwiGLU: PF_CW / 2 tiles of gate (c < PF_CW / 2) and the same of up
  constexpr int HCW = PF_CW / 2;
  int tile[PF_CW];
  bool tile_ok[PF_CW];
  const u32x4_t* wp[PF_CW];
  const uint32_t* szp[PF_CW];
#pragma unroll
  for (int c = 0; c < PF_CW; ++c) {
    const int t = DUAL ? (cb * PF_WN + wn) * HCW + (c % HCW) : (cb * PF_WN + wn) * PF_CW + c;
    tile_ok[c] = t < a.NTILES;
    tile[c] = min(t, a.NTILES - 1);
    const bool second = DUAL && c >= HCW;
    wp[c] = (second ? a.w1 : a.w0) + (size_t)tile[c] * a.KT * 64 + lane;
    szp[c] = (second ? a.sz1 : a.sz0) + (size_t)tile[c] * a.Gp * 16 + ni;
  }

  // ---- A staging: wave w stages row tile w (all KSTEPS fragments of the k-tile): lane (kb, r) <- 16 bytes of row m0 + 16 w + r
  const int arow = min(m0 + wave * 16 + ni, a.M - 1);  // rows past M re-read the last row (masked on store)
  const char* xrow = reinterpret_cast<const char*>(a.x) + ((size_t)arow * a.ldx + kb * 8) * 2;
  u32x4_t areg[KSTEPS];
  u32x4_t wreg[PF_CW];
  uint32_t szreg[PF_CW] = {};
  auto load_a = [&](int kt) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) areg[ks] = *reinterpret_cast<const u32x4_t*>(xrow + ((size_t)kt * KTILE + ks * 32) * 2);
  };
  auto load_b = [&](int kt) {
#pragma unroll
    for (int c = 0; c < PF_CW; ++c) wreg[c] = __builtin_nontemporal_load(wp[c] + (size_t)kt * 64);
  };
  // stages the loaded k-tile into A buffer `buf`; its row sums go into (or are added to) the row-sum table of its quantisation
  // group, `gbuf` = group parity (the lane with kb == 0 of a row's 4 k-block lanes owns that row's entry: no race)
  auto stage_a = [&](int buf, int gbuf, bool group_start) {
    unsigned char* dst = smem + buf * ABYTES + ((size_t)wave * KSTEPS * 64 + lane) * 16;
    // A = the staged fragment, B = ones: every column of the result tile is the row sum (lane (kb, ni): rows kb * 4 + r)
    const u32x4_t ones = FT == DIHIP_BF16 ? u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}
                                    : u32x4_t{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
    f32x4_t sx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      *reinterpret_cast<u32x4_t*>(dst + ks * 1024) = areg[ks];
      sx = mfma16<FT>(areg[ks], ones, sx);
    }
    if (ni == 0) {
      f32x4_t* p = reinterpret_cast<f32x4_t*>(xsum + gbuf * PF_BM + wave * 16 + kb * 4);
      if (group_start) {
        *p = sx;
      } else {
        const f32x4_t o = *p;
        *p = f32x4_t{o[0] + sx[0], o[1] + sx[1], o[2] + sx[2], o[3] + sx[3]};
      }
    }
  };

  // acc: exact integer products of the running quantisation group (MFMA); tot: the scaled result
  f32x4_t acc[PF_RT][PF_CW], tot[PF_RT][PF_CW];
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < PF_RT; ++rt)
#pragma unroll
    for (int c = 0; c < PF_CW; ++c) acc[rt][c] = tot[rt][c] = zero4;
  uint32_t ex_mask = 0x000F000Fu, ex_magic = FT == DIHIP_BF16 ? 0x43004300u : 0x64006400u;
  asm volatile("" : "+v"(ex_mask), "+v"(ex_magic));

  // ---- prologue: the first k-tile into buffer 0 ----------------------------------------------------------------
  int gl = 0;                                                   // k-tiles of the running group done
  int grp = GPT ? kt0 : (a.ktpg < a.KT ? kt0 / a.ktpg : 0);     // running group (a K part starts on a group boundary)
  load_a(kt0);
  load_b(kt0);
#pragma unroll
  for (int c = 0; c < PF_CW; ++c) szreg[c] = szp[c][(size_t)min(grp, a.Gp - 1) * 16];
  stage_a(0, grp & 1, true);
  __syncthreads();

#ifdef DIHIP_PF_TRACE
  // k-tiles 8..11 of every workgroup: 8 stamps per k-tile (loop top, after each k-step, after the fix-up, after staging, after the barrier)
#define DIHIP_PF_STAMP(I)                                                                                                \
  do {                                                                                                                   \
    if (a.trace && lane == 0 && kt >= 8 && kt < 12)                                                                       \
      a.trace[(((size_t)blockIdx.x * PF_WAVES + wave) * 4 + (kt - 8)) * 8 + (I)] = __builtin_readcyclecounter();         \
  } while (0)
#else
#define DIHIP_PF_STAMP(I) do { } while (0)
#endif
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    DIHIP_PF_STAMP(0);
    u32x4_t wcur[PF_CW];
#pragma unroll
    for (int c = 0; c < PF_CW; ++c) wcur[c] = wreg[c];
    const bool more = kt + 1 < kt1;
    if (more) {  // (uniform) the loads of the next k-tile fly while this one is multiplied
      load_a(kt + 1);
      load_b(kt + 1);
    }
    // ---- multiply: every wave reads all row tiles of the A tile
    const unsigned char* abase = smem + buf * ABYTES + (size_t)lane * 16;
    __builtin_amdgcn_s_setprio(1);  // the multiplying wave wins the issue arbitration against its SIMD neighbour's VALU phase
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      u32x4_t bf[PF_CW];
#pragma unroll
      for (int c = 0; c < PF_CW; ++c) {
        if constexpr ((DIHIP_PF_X & 8) != 0) bf[c] = wcur[c];
        else bf[c] = EX::frag(wcur[c], ks, ex_mask, ex_magic);
      }
#pragma unroll
      for (int rt = 0; rt < PF_RT; ++rt) {
        const u32x4_t af = *reinterpret_cast<const u32x4_t*>(abase + ((size_t)(wm * PF_RT + rt) * KSTEPS + ks) * 1024);
#pragma unroll
        for (int c = 0; c < PF_CW; ++c) acc[rt][c] = mfma16<FT>(af, bf[c], acc[rt][c]);
      }
#ifdef DIHIP_PF_TRACE
      if (ks == 0) DIHIP_PF_STAMP(1);
      if (ks == 1) DIHIP_PF_STAMP(2);
      if (ks == 2) DIHIP_PF_STAMP(3);
      if (ks == 3) DIHIP_PF_STAMP(4);
#endif
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- at the group's end: scale / zero-point on the f32 accumulators, with the group's row sums
    ++gl;
    const bool gend = (GPT || gl == a.ktpg || !more) && (!(DIHIP_PF_X & 1) || !more);
    if (gend) {  // (uniform)
      const float* xs = xsum + (grp & 1) * PF_BM + wm * PF_RT * 16 + kb * 4;
      float s_[PF_CW], nz_[PF_CW];
#pragma unroll
      for (int c = 0; c < PF_CW; ++c) {
        s_[c] = ft_bits_to_f32<FT>(szreg[c] & 0xFFFFu);
        nz_[c] = -(ft_bits_to_f32<FT>(szreg[c] >> 16) + EX::OFFSET);
      }
#pragma unroll
      for (int rt = 0; rt < PF_RT; ++rt) {
        const f32x4_t xv = *reinterpret_cast<const f32x4_t*>(xs + rt * 16);
#pragma unroll
        for (int c = 0; c < PF_CW; ++c) {
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[rt][c][r] = fmaf(s_[c], fmaf(nz_[c], xv[r], acc[rt][c][r]), tot[rt][c][r]);
          acc[rt][c] = zero4;
        }
      }
      gl = 0;
      ++grp;
      if (more) {
#pragma unroll
        for (int c = 0; c < PF_CW; ++c) szreg[c] = szp[c][(size_t)min(grp, a.Gp - 1) * 16];
      }
    }
    // ---- the next A tile into the other buffer (its last readers passed the previous barrier); its row sums belong to
    // group `grp` (already advanced when this k-tile closed one): that group's table was last read two groups ago
    DIHIP_PF_STAMP(5);
    if (more && !(DIHIP_PF_X & 2)) stage_a(buf ^ 1, grp & 1, gl == 0);
    DIHIP_PF_STAMP(6);
    if constexpr (!(DIHIP_PF_X & 4)) __syncthreads();
    DIHIP_PF_STAMP(7);
  }
#undef DIHIP_PF_STAMP

  // ---- tail split: this part's f32 tile to the slab (rows / columns of the workgroup tile; SwiGLU: gate columns, then up) ----
  if (split_part >= 0) {  // (uniform)
    float* st = a.slab + ((size_t)split_tile * a.ksplit + split_part) * (PF_BM * 256);
#pragma unroll
    for (int rt = 0; rt < PF_RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (wm * PF_RT + rt) * 16 + kb * 4 + r;
#pragma unroll
        for (int c = 0; c < PF_CW; ++c) {
          const int col = DUAL ? (c < HCW ? 0 : 128) + (wn * HCW + (c % HCW)) * 16 + ni : (wn * PF_CW + c) * 16 + ni;
          st[(size_t)row * 256 + col] = tot[rt][c][r];
        }
      }
    return;
  }
  // ---- epilogue: lane (kb, ni) holds rows rt * 16 + kb * 4 + r of column tile c, column ni --------------------
#pragma unroll
  for (int rt = 0; rt < PF_RT; ++rt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + (wm * PF_RT + rt) * 16 + kb * 4 + r;
      if (m >= a.M) continue;
      if constexpr (DUAL) {
#pragma unroll
        for (int c = 0; c < HCW; ++c) {
          const int n = tile[c] * 16 + ni;
          if (tile_ok[c] && n < a.N) {
            const float g = tot[rt][c][r], u = tot[rt][HCW + c][r];
            store_ft<FT>(a.y, (size_t)m * a.ldy + n, (g / (1.f + expf(-g))) * u);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < PF_CW; ++c) {
          const int n = tile[c] * 16 + ni;
          if (!tile_ok[c] || n >= a.N) continue;
          float v = tot[rt][c][r];
          if constexpr (EPI == EPI_STD) {
            v = a.alpha * v;
            if (a.bias) v += load_ft<FT>(a.bias, n);
            v = apply_act(v, a.act);
            if (a.residual) v = ft_round<FT>(v) + load_ft<FT>(a.residual, (size_t)m * a.ldy + n);
            store_ft<FT>(a.y, (size_t)m * a.ldy + n, v);
          } else {
            const float base = a.h_res ? a.h_res[(size_t)m * a.N + n] : 0.f;
            a.h_out[(size_t)m * a.N + n] = base + a.alpha * v;
          }
        }
      }
    }
  }
}

// tail split: the K parts of every split tile added in order + the epilogue of gemm_prefill_kernel
template <int FT, int EPI>
__global__ __launch_bounds__(256) void gemm_prefill_tail_reduce_kernel(const PrefillArgs a) {
  constexpr bool DUAL = EPI == EPI_SWIGLU;
  constexpr int NC = DUAL ? 128 : 256;  // output columns of a workgroup tile
  const int mblocks = (a.M + PF_BM - 1) / PF_BM;
  const size_t total = (size_t)(a.col_blocks - a.tail_cb) * mblocks * PF_BM * NC;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int col = (int)(e % NC);
    const size_t q = e / NC;
    const int row = (int)(q % PF_BM), tile = (int)(q / PF_BM);
    const int cbt = tile / mblocks, mb = tile - cbt * mblocks;
    const int m = mb * PF_BM + row, n = (a.tail_cb + cbt) * NC + col;
    if (m >= a.M || n >= a.N) continue;
    const float* st = a.slab + (size_t)tile * a.ksplit * (PF_BM * 256) + (size_t)row * 256 + col;
    float v = 0.f, v2 = 0.f;
    for (int s = 0; s < a.ksplit; ++s) {
      v += st[(size_t)s * (PF_BM * 256)];
      if constexpr (DUAL) v2 += st[(size_t)s * (PF_BM * 256) + 128];
    }
    if constexpr (DUAL) {
      store_ft<FT>(a.y, (size_t)m * a.ldy + n, (v / (1.f + expf(-v))) * v2);
    } else if constexpr (EPI == EPI_STD) {
      v = a.alpha * v;
      if (a.bias) v += load_ft<FT>(a.bias, n);
      v = apply_act(v, a.act);
      if (a.residual) v = ft_round<FT>(v) + load_ft<FT>(a.residual, (size_t)m * a.ldy + n);
      store_ft<FT>(a.y, (size_t)m * a.ldy + n, v);
    } else {
      const float base = a.h_res ? a.h_res[(size_t)m * a.N + n] : 0.f;
      a.h_out[(size_t)m * a.N + n] = base + a.alpha * v;
    }
  }
}

template <int WBITS, int FT, int EPI, int GPT>
hipError_t launch_gemm_prefill(const PrefillArgs& a, int blocks, hipStream_t stream);

#define DIHIP_DEFINE_PREFILL_LAUNCH(WBITS, FT, EPI, GPT)                                                           \
  template <>                                                                                                      \
  hipError_t launch_gemm_prefill<WBITS, FT, EPI, GPT>(const PrefillArgs& a, int blocks, hipStream_t s) {           \
    auto kern = gemm_prefill_kernel<WBITS, FT, EPI, GPT>;                                                          \
    constexpr size_t lds = prefill_lds_bytes<WBITS>();                                                             \
    if (lds > 64 * 1024) {  /* the grant is per DEVICE: once per device of this process */           \
      static std::atomic<unsigned> granted_mask[4];  /* 128 device ids */                                          \
      int dev = 0;                                                                                                 \
      if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();                                              \
      const unsigned bit = 1u << (dev & 31);                                                                       \
      std::atomic<unsigned>* word = dev >= 0 && dev < 128 ? &granted_mask[dev >> 5] : nullptr;                     \
      if (!word || !(word->load(std::memory_order_acquire) & bit)) {                                               \
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
        if (e != hipSuccess) return e;                                                                             \
        if (word) word->fetch_or(bit, std::memory_order_release);                                                  \
      }                                                                                                            \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(PF_THREADS), lds, s, a);                                           \
    if (a.tail_cb < a.col_blocks) {                                                                                \
      const size_t outs = (size_t)(a.col_blocks - a.tail_cb) * ((a.M + PF_BM - 1) / PF_BM) * PF_BM * 256;          \
      hipLaunchKernelGGL((gemm_prefill_tail_reduce_kernel<FT, EPI>), dim3((unsigned)std::min<size_t>((outs + 1023) / 1024, 2048)), dim3(256), 0, s, a); \
    }                                                                                                              \
    return hipGetLastError();                                                                                      \
  }

#define DIHIP_DEFINE_PREFILL_LAUNCH_SET(WBITS, FT)          \
  DIHIP_DEFINE_PREFILL_LAUNCH(WBITS, FT, EPI_STD, 0)        \
  DIHIP_DEFINE_PREFILL_LAUNCH(WBITS, FT, EPI_STD, 1)        \
  DIHIP_DEFINE_PREFILL_LAUNCH(WBITS, FT, EPI_SWIGLU, 0)     \
  DIHIP_DEFINE_PREFILL_LAUNCH(WBITS, FT, EPI_SWIGLU, 1)     \
  DIHIP_DEFINE_PREFILL_LAUNCH(WBITS, FT, EPI_ADDTO, 0)      \
  DIHIP_DEFINE_PREFILL_LAUNCH(WBITS, FT, EPI_ADDTO, 1)

}  // namespace dihip
