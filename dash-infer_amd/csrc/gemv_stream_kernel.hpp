// gemv_stream_kernel.hpp -- decode-shaped (M <= 16 rows) weight-only GEMV for gfx950.
//
// The batch-1 decode step is a chain of short weight-streaming launches (7-72 MB each).  At
// that size the critical path of ONE launch -- not the sustained rate -- decides the achieved
// HBM GB/s: every dependent global round trip (fence, counter, slab re-read) costs 1-3 us
// against a 1-12 us transfer, and so does every microsecond of prologue during which no load
// is in flight.  This kernel therefore
//
//   * has NO inter-workgroup communication: a workgroup (8 waves) owns whole 16-column tiles;
//     its waves split K (and the tiles) among themselves and combine through LDS in a fixed
//     order (bit-reproducible, no atomics, no fences, no slabs in HBM, no second launch);
//   * streams each wave's flat sequence of 1 KiB weight chunks ("dihip tile-major": one 16-byte
//     load per lane is one MFMA B fragment x KSTEPS) through a register ring of D chunks that is
//     filled right after the activation loads have been issued, with hand-counted s_waitcnt so
//     that D-1 chunks stay in flight while one is consumed;
//   * keeps the code path short (one copy of the consume/refill body; the tail is handled by
//     dummy refills that keep the vmcnt arithmetic uniform) -- six such kernels alternate every
//     few microseconds and share the instruction cache;
//   * expands integers with a one-op magic-number trick, feeds them to the MFMA as exact small
//     numbers and applies scale / zero-point once per quantisation group on the f32
//     accumulator, with sum_k x[k] taken from a per-k-tile table computed once per workgroup;
//   * fuses the epilogues (bias, activation, residual | SwiGLU of a gate/up pair | f32
//     hidden-stream update) and the RMSNorm prologue.
//
// Replaces, for decode shapes, gemv_a16w8_subc_splitk_m1_kernel + reduce_sum
// (gemm_a16w8_subc_kernel.cu:953-1119), the Ampere per-channel split-K kernel
// (gemm_a16w8_perc_kernel.cu:1573-1762) and hgemm_a16w4_subc_32x256x32 (gemm_a16w4_subc_kernel.cu:466-815).
//
// Host-side contract (gemm_lowp.hip: run_gemm / make_gemv_plan): M <= 16, K % KTILE == 0,
// ldx % 8 == 0, x and gamma 16-byte aligned, group_size % KTILE == 0 or per-channel, LDS budget.
#pragma once
#include <atomic>
#include "gemm_lowp_kernel.hpp"

namespace dihip {

constexpr int GEMV_WAVES = 8;
constexpr int GEMV_THREADS = GEMV_WAVES * 64;
constexpr int GEMV_RING = 8;  // 1 KiB chunks in flight per wave
// Slots of the ring filled BEFORE the activation prologue; the rest are requested when the prologue's last barrier is
// reached.  A CU holds ~32 KiB of outstanding misses: 8 waves x 8 chunks asked for at once is twice that, and the second
// half of the workgroup's waves sat in the issue queue ~0.8 us behind the first (profiles/r02_gemv_wave_timeline.txt: ring
// issued at 1.5 vs 2.3 us), reached the prologue's barriers late and finished the main loop ~1.1 us after waves 0-3.
// 4 + 4 asks for exactly the capacity up front and tops the ring up once the activations are staged.
#ifndef DIHIP_GEMV_RING_EARLY
#define DIHIP_GEMV_RING_EARLY 4
#endif
constexpr int GEMV_RING_EARLY = DIHIP_GEMV_RING_EARLY;

static_assert(GEMV_RING_EARLY >= 1 && GEMV_RING_EARLY <= GEMV_RING, "early slots: 1 .. ring");

// ---- hand-scheduled weight stream ------------------------------------------------------------
// The ring loads are issued through inline asm so that hipcc's waitcnt pass does not see them:
// with control flow in the loop it falls back to draining the ring every iteration (vmcnt(1)
// at the loop head).  The loop places its own counted `s_waitcnt vmcnt(N)` instead
// (cdna_hip_programming.md section 5.7 / T3+T4).  saddr form: wave-uniform 64-bit base in
// SGPRs + a 32-bit per-lane byte offset.
__device__ __forceinline__ void stream_load_b128(u32x4_t& dst, const void* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(dst) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void stream_load_b32(uint32_t& dst, const void* sbase, uint32_t voff) {
  asm volatile("global_load_dword %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase));
}
// a second load into the SAME register (read-write operand: the register's current value is not dead)
__device__ __forceinline__ void stream_reload_b32(uint32_t& dst, const void* sbase, uint32_t voff) {
  asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void stream_load_plain_b128(u32x4_t& dst, const void* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase));
}
// pins a wave-uniform pointer into SGPRs (folds away when it already lives there): the ring loads take their base
// through an "s" constraint, and hipcc occasionally carries a uniform address computation on the VALU
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {  // DPP move, all rows / banks enabled
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int N>
__device__ __forceinline__ void stream_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
}
// ties registers to the preceding wait: their uses cannot be scheduled above the s_waitcnt
__device__ __forceinline__ void stream_landed(u32x4_t& w, uint32_t& s) { asm volatile("" : "+v"(w), "+v"(s)); }
// Activation loads of the prologue are issued BEFORE the ring fill and waited for with a counted
// vmcnt (loads return in order: a compiler-visible load issued after the ring would only be
// usable once the whole ring has landed, serialising prologue and first HBM round trip).
__device__ __forceinline__ void early_landed(u32x4_t& v) { asm volatile("" : "+v"(v)); }


// W4 expansion for the GEMV: pair I of a dword is (d >> 4I) & 0x000F000F; OR-ing the exponent of
// 128.0 (bf16) gives OFFSET + q exactly in both halves.  mask / magic are passed as
// (opaque) registers so that each pair is one v_lshrrev + one v_and_or_b32.
template <int WBITS, int FT>
struct ExpandV {
  static constexpr float OFFSET = Expand<WBITS, FT>::OFFSET;
  __device__ __forceinline__ static u32x4_t frag(const u32x4_t& chunk, int ks, uint32_t, uint32_t) {
    return Expand<WBITS, FT>::frag(chunk, ks);
  }
};
// bf16 only: for f16 the 1024 offset would cost 7 bits of the f32 accumulator against an 11-bit
// output; f16 keeps the offset-16 expansion of the general kernel (Expand<4, DIHIP_F16>).
template <>
struct ExpandV<4, DIHIP_BF16> {
  static constexpr float OFFSET = 128.f;
  __device__ __forceinline__ static u32x4_t frag(const u32x4_t& chunk, int ks, uint32_t mask, uint32_t magic) {
    const uint32_t d = chunk[ks];
    return u32x4_t{(d & mask) | magic, ((d >> 4) & mask) | magic, ((d >> 8) & mask) | magic, ((d >> 12) & mask) | magic};
  }
};

struct GemvArgs {
  const u32x4_t* w0;
  const u32x4_t* w1;    // EPI_SWIGLU: "up" weight
  const uint32_t* sz0;  // [NTILES][Gp][16] (scale | zero << 16)
  const uint32_t* sz1;
  const void* x;  // PRO_PLAIN: FT [M, ldx]; PRO_RMSNORM: f32 [M, ldx]
  int ldx;
  const void* gamma;
  float eps;
  const void* bias;
  const void* residual;
  void* y;
  int ldy;
  const float* h_res;
  float* h_out;
  float alpha;
  int act;
  int M, N, K;
  int KT;      // k-tiles of the packed weight
  int NTILES;  // 16-column tiles
  int Gp;      // (scale, zero) groups stored per tile
  int ktpg;    // k-tiles per quantisation group (per-channel: >= KT)
  int kgroups; // K-split units: quantisation groups (sub-channel) or k-tiles (per-channel / W16)
  int upb;     // units (column tiles; tile pairs for SwiGLU) per workgroup
  int nu_q, nu_r;  // NTILES / blocks, NTILES % blocks (host): block b owns nu_q + (b < nu_r) units
  // SLOT variant only (mixture-of-experts, M = 1 per slot)
  const int* slot_expert;        // [gridDim.y] expert of the slot, < 0: skip
  size_t w_estride, sz_estride;  // u32x4 / u32 elements between consecutive experts' packed tensors
  int x_div;                     // activation row of slot s = s / x_div
  int nslots;
  // SLOT with MR > 1 (grouped slots): gridDim.y enumerates GROUPS of up to 4 slots that picked the same expert; group g
  // streams expert slot_expert[g] once for its slot_nrows[g] rows, reads activation rows slot_rows[4g + r] / x_div and
  // writes rows slot_rows[4g + r]
  const int* slot_rows;
  const int* slot_nrows;
  int WK, WN;  // wave grid inside the workgroup, WK * WN == GEMV_WAVES, both powers of two (SwiGLU: WN >= 2)
  // K-slice boundaries in split units (k-groups), kcut[0] = 0 .. kcut[WK] = kgroups: host-computed, so that the slices need not
  // be equal (the second-dispatched half of a workgroup streams slower: gemm_lowp.hip fill_kcut)
  int kcut[GEMV_WAVES + 1];
  int wmap;    // wave -> (wk, wn): 0: wk = wave % WK, wn = wave / WK;  1: wn = wave % WN, wk = wave / WN (k-slices ordered by wave age)
  int RS;      // LDS activation row stride in elements (KT * KTILE + 8)
  unsigned long long* trace;  // diagnostics (dihip_debug_set_trace): [block][wave][8] wall-clock stamps, or null
  // HAND != 0 only (decode_mlp_block.hip: two GEMVs of ONE launch hand a row over in-launch).  Granules = 8 bytes {two packed FT
  // elements, tag}: the data is the flag (MI355X guide, Guideline 16 R2); `hand_flags[b] = tag` once producer workgroup b's granules have
  // drained, so that consumers poll nproducers words instead of sweeping the row until it is complete.  tag = the launch's epoch.
  unsigned long long* hand_gran;  // [N / 2] (HAND 1: written; HAND 2: the activation row, K / 2 granules)
  unsigned* hand_flags;           // [hand_nproducers]
  unsigned* hand_err;             // set non-zero when a bounded wait gave up
  unsigned hand_spin_limit;  // (the tag is the launch's epoch, read on the device: a parameter of the body)
  int hand_nproducers;
};

// LDS carve-up (bytes): [zero block 256][xs : rows * RS * 2][xsum : KT * 16 * 4][red : upb*DUAL * WK * rows * 16 * 4]
__host__ __device__ inline size_t gemv_xs_bytes(int rows, int RS) { return ((size_t)rows * RS * 2 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t gemv_lds_bytes(int rows, int RS, int KT, int upb, int dual, int WK) {
  size_t red = (size_t)upb * (dual ? 2 : 1) * WK * rows * 16 * 4;
  if (red < 1024) red = 1024;  // also holds the RMSNorm partial sums
  return 256 + gemv_xs_bytes(rows, RS) + (size_t)KT * 16 * 4 + red;
}

// GPT: every k-tile is one quantisation group (W4 g128, W8 g64): no accumulator carry between chunks
// SLOT (mixture-of-experts): gridDim.y enumerates (token, expert-rank) slots; slot s streams the weights of expert
// slot_expert[s] (all experts have one shape: base + expert * stride), reads activation row s / x_div and writes row s.
// HAND (decode_mlp_block.hip, M = 1): 1 -- EPI_SWIGLU publishes its outputs as granules + this workgroup's flag instead of storing
// FT elements; 2 -- PRO_PLAIN takes its activation row from granules: the WHOLE ring is requested first (the weights do not depend
// on the row: they stream while the producers finish), then the producers' flags are awaited and the row is swept into LDS.
template <int WBITS, int FT, int MR, int PRO, int EPI, int GPT, bool SLOT = false, int HAND = 0>
__device__ __forceinline__ void gemv_stream_body(const GemvArgs& a, const int bid, const int nblocks, unsigned char* smem,
                                                 const unsigned hand_tag = 0u) {
  static_assert(HAND == 0 || (MR == 1 && !SLOT && ((HAND == 1 && EPI == EPI_SWIGLU) || (HAND == 2 && PRO == PRO_PLAIN))),
                "in-launch hand-off: one row, SwiGLU producer / plain-prologue consumer");
  // the tensors of this launch (SLOT: of this slot's expert / activation row); everything else is read from `a`
  const u32x4_t* p_w0 = a.w0;
  const u32x4_t* p_w1 = a.w1;
  const uint32_t* p_sz0 = a.sz0;
  const uint32_t* p_sz1 = a.sz1;
  const void* p_x = a.x;
  void* p_y = a.y;
  if constexpr (SLOT) {
    const int s_ = blockIdx.y;
    const int e_ = a.slot_expert[s_];
    if (e_ < 0) return;  // expert not on this rank (expert parallelism): the slot contributes nothing
    p_w0 = a.w0 + (size_t)e_ * a.w_estride;
    p_sz0 = a.sz0 + (size_t)e_ * a.sz_estride;
    if (a.w1) p_w1 = a.w1 + (size_t)e_ * a.w_estride;
    if (a.sz1) p_sz1 = a.sz1 + (size_t)e_ * a.sz_estride;
    if constexpr (MR == 1) {
      p_x = reinterpret_cast<const uint16_t*>(a.x) + (size_t)(s_ / a.x_div) * a.ldx;
      p_y = reinterpret_cast<uint16_t*>(a.y) + (size_t)s_ * a.ldy;
    } else {
      // row 0 of the group: the early activation loads take their base from p_x (rows > 0: x_row below)
      const int r0_ = __builtin_amdgcn_readfirstlane(a.slot_rows[(size_t)s_ * 4]);
      p_x = reinterpret_cast<const uint16_t*>(a.x) + (size_t)(r0_ / a.x_div) * a.ldx;
    }
  }
  // activation row r of this workgroup's rows (wave-uniform: feeds the "s" base of the asm loads)
  auto x_row = [&](int r) -> const uint16_t* {
    if constexpr (SLOT && MR > 1) {
      const int sr_ = __builtin_amdgcn_readfirstlane(a.slot_rows[(size_t)blockIdx.y * 4 + r]);
      return reinterpret_cast<const uint16_t*>(a.x) + (size_t)(sr_ / a.x_div) * a.ldx;
    } else {
      return reinterpret_cast<const uint16_t*>(p_x) + (size_t)r * a.ldx;
    }
  };
  using WT = WTraits<WBITS>;
  using EX = ExpandV<WBITS, FT>;
  constexpr int KSTEPS = WT::KSTEPS;
  constexpr int KTILE = WT::KTILE;
  constexpr bool QUANT = WBITS != 16;
  constexpr int DUAL = EPI == EPI_SWIGLU ? 2 : 1;
  constexpr int D = GEMV_RING;
  constexpr int LPC = QUANT ? 2 : 1;  // loads per chunk

  const int rows = MR == 1 ? 1 : (SLOT ? __builtin_amdgcn_readfirstlane(a.slot_nrows[blockIdx.y]) : a.M);  // <= 16
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem + 256);
  float* xsum_tab = reinterpret_cast<float*>(smem + 256 + gemv_xs_bytes(rows, a.RS));
  float* red = xsum_tab + (size_t)a.KT * 16;

  const int tid = threadIdx.x;
  // wave-uniform bookkeeping lives in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int ni = lane & 15, kb = lane >> 4;
  const int lgWK = __builtin_ctz(a.WK), lgWN = __builtin_ctz(a.WN);
  const int wk = a.wmap ? wave >> lgWN : wave & (a.WK - 1), wn = a.wmap ? wave & (a.WN - 1) : wave >> lgWK;

  // per-wave wall-clock stamps (tools/gemv_bench TRACE=1): compiled in only with -DDIHIP_GEMV_TRACE=1 (`make trace` ->
  // lib/trace/libdashinfer_hip.so).  In the product build they are absent: each one is an exec-masked branch that splits
  // the scalar-heavy ramp into basic blocks and pins the kernel-argument loads behind it.
#if defined(DIHIP_GEMV_TRACE) && DIHIP_GEMV_TRACE
#define DIHIP_GEMV_STAMP(I)                                                                        \
  do {                                                                                             \
    /* every lane stores the same value to the same slot: a wave-uniform branch only (a divergent `lane == 0` branch */ \
    /* makes hipcc carry the ring pointers through VGPRs across it, which the "s" operands of the asm loads reject) */  \
    if (a.trace) a.trace[((size_t)bid * GEMV_WAVES + wave) * 8 + (I)] = wall_clock64();                           \
  } while (0)
#else
#define DIHIP_GEMV_STAMP(I) do { } while (0)
#endif
  DIHIP_GEMV_STAMP(0);

  // ---- early activation loads -------------------------------------------------------------------
  // Every thread owns 8-element vectors i = j*THREADS + tid of a row (k = 8i): PRO_PLAIN loads 16
  // bytes of x, PRO_RMSNORM 2 x 16 bytes of the f32 hidden stream + 16 bytes of gamma.  Vector
  // indices beyond the row are clamped (harmless reload) instead of predicated.
  // `ev` / `eg` hold ONLY the batch issued here, ahead of the ring fill (row 0, vectors 0 ..): written by this asm, read after
  // the wait in the prologue, never assigned again.  (They used to be re-loaded for the later rows: two definitions make a
  // phi, the compiler resolves a phi with register copies wherever it likes -- it put them BEFORE the wait, and row 0 of a
  // batch was now and then read before it had landed.  tools/audit_asm_loads.py checks the build for this.)
  // Every later batch (rows > 0, rows longer than one batch) goes through fetch_batch into registers of its own.
  const int Kp = a.KT * KTILE;  // == K (host contract)
  const int nvec = Kp >> 3;
  constexpr int NE = PRO == PRO_RMSNORM ? 2 : 8;  // vectors per thread per batch
  constexpr int NV = PRO == PRO_RMSNORM ? 2 * NE : NE, NG = PRO == PRO_RMSNORM ? NE : 1;
  u32x4_t ev[NV];
  u32x4_t eg[NG];
  if constexpr (HAND != 2) {
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      // sub-batches entirely beyond the row are skipped (wave-uniform); a partial one is clamped
      if (j > 0 && j * GEMV_THREADS >= nvec) {
        ev[PRO == PRO_RMSNORM ? 2 * j : j] = u32x4_t{0u, 0u, 0u, 0u};
        if constexpr (PRO == PRO_RMSNORM) {
          ev[2 * j + 1] = u32x4_t{0u, 0u, 0u, 0u};
          eg[j] = u32x4_t{0u, 0u, 0u, 0u};
        }
        continue;
      }
      const uint32_t i = (uint32_t)min(j * GEMV_THREADS + tid, nvec - 1);
      if constexpr (PRO == PRO_RMSNORM) {
        stream_load_plain_b128(ev[2 * j], p_x, i * 32u);
        stream_load_plain_b128(ev[2 * j + 1], p_x, i * 32u + 16u);
        stream_load_plain_b128(eg[j], a.gamma, i * 16u);
      } else {
        stream_load_plain_b128(ev[j], p_x, i * 16u);
      }
    }
  }
  // (asm loads as well, waited for on the spot: loads the compiler tracks itself make it guard every later re-use of their
  // registers with `s_waitcnt vmcnt(0)` -- also on the paths that never issued them, which drained the ring before the main loop)
  auto fetch_batch = [&](int r, int v0, u32x4_t (&V)[NV], u32x4_t (&G)[NG]) {
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      if (j > 0 && v0 + j * GEMV_THREADS >= nvec) {
        V[PRO == PRO_RMSNORM ? 2 * j : j] = u32x4_t{0u, 0u, 0u, 0u};
        if constexpr (PRO == PRO_RMSNORM) {
          V[2 * j + 1] = u32x4_t{0u, 0u, 0u, 0u};
          G[j] = u32x4_t{0u, 0u, 0u, 0u};
        }
        continue;
      }
      const uint32_t i = (uint32_t)min(v0 + j * GEMV_THREADS + tid, nvec - 1);
      if constexpr (PRO == PRO_RMSNORM) {
        const float* hrow = reinterpret_cast<const float*>(p_x) + (size_t)r * a.ldx;
        stream_load_plain_b128(V[2 * j], hrow, i * 32u);
        stream_load_plain_b128(V[2 * j + 1], hrow, i * 32u + 16u);
        stream_load_plain_b128(G[j], a.gamma, i * 16u);
      } else {
        stream_load_plain_b128(V[j], x_row(r), i * 16u);
      }
    }
    stream_wait<0>();
#pragma unroll
    for (int j = 0; j < NV; ++j) early_landed(V[j]);
#pragma unroll
    for (int j = 0; j < NG; ++j) early_landed(G[j]);
  };
  DIHIP_GEMV_STAMP(7);

  // ---- this workgroup's units and this wave's share ---------------------------------------
  // units are dealt round-robin: workgroup b owns units b, b + NB, b + 2*NB, ... (SwiGLU: (gate, up) tile
  // pairs), which spreads every workgroup's address range over the whole matrix
  const int NB = nblocks;
  const int u0 = __builtin_amdgcn_readfirstlane(bid);  // pinned scalar: see uniform_ptr
  const int nu = a.nu_q + (u0 < a.nu_r ? 1 : 0);  // (NTILES - u0 + NB - 1) / NB without a device-side division
  const int nv = nu * DUAL;                  // half-units (one weight tile each)
  // K split in whole quantisation groups (per-channel: any k-tile boundary)
  const bool subc = QUANT && a.ktpg < a.KT;
  const int gsz = subc ? a.ktpg : 1;
  const int gcount = subc ? a.ktpg : (1 << 30);  // group countdown start (per-channel: never expires)
  const int g_lo = __builtin_amdgcn_readfirstlane(a.kcut[wk]);
  const int k_lo = min(a.KT, g_lo * gsz);
  const int k_hi = min(a.KT, __builtin_amdgcn_readfirstlane(a.kcut[wk + 1]) * gsz);
  const int nk = k_hi - k_lo;
  const int nvw = wn < nv ? (nv - wn + a.WN - 1) >> lgWN : 0;  // half-units v = wn, wn + WN, ...
  const int total = __builtin_amdgcn_readfirstlane(nvw * nk);

  // ---- register ring ----------------------------------------------------------------------
  u32x4_t wb[D];
  uint32_t sb[D];
#pragma unroll
  for (int j = 0; j < D; ++j) sb[j] = 0u;
  // A wave walks tiles t0, t0 + tstep, ... of ONE matrix (SwiGLU: WN is even, so the parity of the
  // half-unit index -- gate or up -- is fixed per wave); chunk pointers advance by constant strides.
  // The refill is branch-free scalar code; once the sequence is exhausted the cursor parks on a
  // dummy chunk (the head of the matrix), which keeps the vmcnt arithmetic uniform in the tail.
  const bool second = DUAL == 2 && (wn & 1);
  const int t0 = u0 + (DUAL == 2 ? wn >> 1 : wn) * NB;
  const int tstep = (DUAL == 2 ? a.WN >> 1 : a.WN) * NB;
  const char* const dummy_w = reinterpret_cast<const char*>(p_w0);
  const char* const dummy_s = dummy_w;
  const char* wtile = uniform_ptr(reinterpret_cast<const char*>((second ? p_w1 : p_w0) + ((size_t)t0 * a.KT + k_lo) * 64));
  const char* stile = QUANT ? uniform_ptr(reinterpret_cast<const char*>((second ? p_sz1 : p_sz0) + ((size_t)t0 * a.Gp + (subc ? g_lo : 0)) * 16))
                            : dummy_s;
  const size_t wstep = (size_t)tstep * a.KT * 1024, sstep = QUANT ? (size_t)tstep * a.Gp * 64 : 0;
  int to_issue = total;  // real chunks not yet issued
  const char* iwp = wtile;
  const char* isp = stile;
  int ikt = k_lo, igl = gcount;  // issue cursor: k-tile, group countdown
  const uint32_t voff_w = (uint32_t)lane * 16u, voff_s = (uint32_t)ni * 4u;
  // next real chunk into ring slot SLOT
#define DIHIP_GEMV_ISSUE(SLOT)                                       \
  do {                                                               \
    stream_load_b128(wb[SLOT], iwp, voff_w);                         \
    if constexpr (QUANT) stream_load_b32(sb[SLOT], isp, voff_s);     \
    iwp += 1024;                                                     \
    if constexpr (QUANT && GPT) {                                    \
      isp += 64;                                                     \
    } else if constexpr (QUANT) {                                    \
      if (--igl == 0) {                                              \
        igl = gcount;                                                \
        isp += 64;                                                   \
      }                                                              \
    }                                                                \
    if (++ikt == k_hi) {                                             \
      ikt = k_lo;                                                    \
      igl = gcount;                                                  \
      wtile += wstep;                                                \
      stile += sstep;                                                \
      iwp = wtile;                                                   \
      isp = stile;                                                   \
    }                                                                \
  } while (0)
  // (tail) dummy loads that keep the vmcnt arithmetic uniform: one L2-resident word each, never used.  Both go into the
  // slot's scale register -- the second one as a read-write operand, so that the first is not a dead definition -- and the
  // slots stay alive up to the final wait below.  (A load into a dead variable gets whatever register is free; the compiler
  // hands that register to the next value that needs one, and the load lands on top of it whenever it arrives: it was the
  // zero the accumulators are reset from.)
#define DIHIP_GEMV_DUMMY(SLOT)                                       \
  do {                                                               \
    if constexpr (QUANT) {                                           \
      stream_load_b32(sb[SLOT], dummy_w, 0u);                        \
      stream_reload_b32(sb[SLOT], dummy_w, 0u);                      \
    } else {                                                         \
      stream_load_b128(wb[SLOT], dummy_w, 0u); /* no scale word */   \
    }                                                                \
  } while (0)
  constexpr int DE = HAND == 2 ? D : GEMV_RING_EARLY;
#pragma unroll
  for (int j = 0; j < DE; ++j) {
    if (to_issue > 0) {
      DIHIP_GEMV_ISSUE(j);
      --to_issue;
    } else {
      DIHIP_GEMV_DUMMY(j);
    }
  }
  DIHIP_GEMV_STAMP(1);

  // ---- activation prologue ------------------------------------------------------------------
  // x (normalised for PRO_RMSNORM) goes to LDS as FT rows; the per-k-tile sums
  // xsum_tab[kt][row] = sum_{k in tile} x[row][k] are taken on the way: 8-element partials in a
  // fixed pairwise order, combined over the KTILE/8 lanes of a tile by an xor butterfly (DPP).
  if (tid < 16) reinterpret_cast<u32x4_t*>(smem)[tid] = u32x4_t{0u, 0u, 0u, 0u};  // zero block
  auto stage_vector = [&](int r, int i, const u32x4_t& v) {  // i < nvec (wave-uniform validity handled by caller)
    *reinterpret_cast<u32x4_t*>(xs + (size_t)r * a.RS + i * 8) = v;
    if constexpr (QUANT) {
      float e[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        e[2 * q] = ft_bits_to_f32<FT>(v[q] & 0xFFFFu);
        e[2 * q + 1] = ft_bits_to_f32<FT>(v[q] >> 16);
      }
      float sum = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
      sum += dpp_f32<0xB1>(sum);                           // quad_perm [1,0,3,2]: xor 1
      sum += dpp_f32<0x4E>(sum);                           // quad_perm [2,3,0,1]: xor 2
      if constexpr (KTILE >= 64) sum += dpp_f32<0x141>(sum);   // row_half_mirror: other quad of the 8
      if constexpr (KTILE >= 128) sum += dpp_f32<0x140>(sum);  // row_mirror: other half of the 16
      if ((i & (KTILE / 8 - 1)) == 0) xsum_tab[(i / (KTILE / 8)) * 16 + r] = sum;
    }
  };
  if constexpr (HAND != 2) {
    stream_wait<DE * LPC>();  // everything older than the (early part of the) ring fill has landed: the early batch
#pragma unroll
    for (int j = 0; j < NV; ++j) early_landed(ev[j]);
#pragma unroll
    for (int j = 0; j < NG; ++j) early_landed(eg[j]);
  }
  DIHIP_GEMV_STAMP(2);
  if constexpr (HAND == 2) {
    // ---- the row arrives from the producer workgroups of this launch: one wave polls their flags (one word each), then every
    // thread sweeps its 8-element vectors (4 granules each; a tag that is not this launch's means "not yet": retry) ----
    if (wave == 0) {
      for (unsigned spins = 0;; ++spins) {
        // (all flag words of a pass in flight together: a loop that tests as it loads is one round trip per 64 producers)
        unsigned fv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          fv[j] = __hip_atomic_load(a.hand_flags + min(j * 64 + lane, a.hand_nproducers - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) ok = ok && fv[j] == hand_tag;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        if (spins > a.hand_spin_limit) {
          if (lane == 0) __hip_atomic_store(a.hand_err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    // two granules per 16-byte load (each granule is whole inside its aligned half), 5 vectors = 10 loads per thread in flight:
    // the 18944-element row of the 7B down projection is ONE round trip per workgroup
    const auto grsrc = __builtin_amdgcn_make_buffer_rsrc(a.hand_gran, 0, (int)((size_t)nvec * 32), 0x00020000);
    constexpr int VB = 5;
    for (int v0 = 0; v0 < nvec; v0 += VB * GEMV_THREADS) {
      u32x4_t gl[VB][2];
      for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int j = 0; j < VB; ++j) {
          const uint32_t off = (uint32_t)min(v0 + j * GEMV_THREADS + tid, nvec - 1) * 32u;
          gl[j][0] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off, 0, 16 /* sc1 */);
          gl[j][1] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, off + 16u, 0, 16);
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < VB; ++j)
          ok = ok && gl[j][0][1] == hand_tag && gl[j][0][3] == hand_tag && gl[j][1][1] == hand_tag && gl[j][1][3] == hand_tag;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        if (spins > a.hand_spin_limit) {
          if (lane == 0) __hip_atomic_store(a.hand_err, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
#pragma unroll
      for (int j = 0; j < VB; ++j) {
        // (whole 16-lane rows are valid or not together: nvec is a multiple of KTILE / 8)
        const int i = v0 + j * GEMV_THREADS + tid;
        if (i < nvec) stage_vector(0, i, u32x4_t{gl[j][0][0], gl[j][0][2], gl[j][1][0], gl[j][1][2]});
      }
    }
  } else
  if constexpr (PRO == PRO_RMSNORM) {
    // LayerNormNoBeta of the f32 hidden stream (csrc/core/kernel/cpu/layernorm.cpp:110-157):
    // rstd = 1/sqrt(mean(x^2)+eps); x_norm = FT((gamma*x)*rstd)
    auto sumsq = [&](int v0, const u32x4_t (&V)[NV], float ss) {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (v0 + j * GEMV_THREADS + tid < nvec) {
          const f32x4_t h0 = __builtin_bit_cast(f32x4_t, V[2 * j]), h1 = __builtin_bit_cast(f32x4_t, V[2 * j + 1]);
#pragma unroll
          for (int q = 0; q < 4; ++q) ss = fmaf(h0[q], h0[q], ss);
#pragma unroll
          for (int q = 0; q < 4; ++q) ss = fmaf(h1[q], h1[q], ss);
        }
      }
      return ss;
    };
    auto norm_stage = [&](int r, int v0, const u32x4_t (&V)[NV], const u32x4_t (&G)[NG], float rstd) {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int i = v0 + j * GEMV_THREADS + tid;
        if (i < nvec) {
          const f32x4_t h0 = __builtin_bit_cast(f32x4_t, V[2 * j]), h1 = __builtin_bit_cast(f32x4_t, V[2 * j + 1]);
          u32x4_t o;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float ga = ft_bits_to_f32<FT>(G[j][q] & 0xFFFFu), gb = ft_bits_to_f32<FT>(G[j][q] >> 16);
            const float xa = q < 2 ? h0[2 * q] : h1[2 * q - 4], xb = q < 2 ? h0[2 * q + 1] : h1[2 * q - 3];
            o[q] = pack_ft2<FT>((ga * xa) * rstd, (gb * xb) * rstd);  // (one packed convert: same round-to-nearest-even bits)
          }
          stage_vector(r, i, o);
        }
      }
    };
    // V / G: the first batch of row r, landed; later batches of a long row are fetched (twice: sum, then normalise)
    auto rms_row = [&](int r, const u32x4_t (&V)[NV], const u32x4_t (&G)[NG]) {
      u32x4_t tv[NV], tg[NG];
      float ss = sumsq(0, V, 0.f);
      for (int v0 = NE * GEMV_THREADS; v0 < nvec; v0 += NE * GEMV_THREADS) {
        fetch_batch(r, v0, tv, tg);
        ss = sumsq(v0, tv, ss);
      }
      ss = wave_sum(ss);
      if (r > 0) __syncthreads();  // previous row's readers of `red` are done
      if (lane == 0) red[wave] = ss;
      __syncthreads();
      float tot_ss = 0.f;
#pragma unroll
      for (int w = 0; w < GEMV_WAVES; ++w) tot_ss += red[w];
      const float rstd = 1.f / sqrtf(tot_ss / (float)a.K + a.eps);
      norm_stage(r, 0, V, G, rstd);
      for (int v0 = NE * GEMV_THREADS; v0 < nvec; v0 += NE * GEMV_THREADS) {
        fetch_batch(r, v0, tv, tg);
        norm_stage(r, v0, tv, tg, rstd);
      }
    };
    rms_row(0, ev, eg);
    for (int r = 1; r < rows; ++r) {
      u32x4_t lv[NV], lg[NG];
      fetch_batch(r, 0, lv, lg);
      rms_row(r, lv, lg);
    }
  } else {
    auto plain_row = [&](int r, const u32x4_t (&V)[NV]) {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int i = j * GEMV_THREADS + tid;
        if (i < nvec) stage_vector(r, i, V[j]);
      }
      for (int v0 = NE * GEMV_THREADS; v0 < nvec; v0 += NE * GEMV_THREADS) {
        u32x4_t tv[NV], tg[NG];
        fetch_batch(r, v0, tv, tg);
#pragma unroll
        for (int j = 0; j < NE; ++j) {
          const int i = v0 + j * GEMV_THREADS + tid;
          if (i < nvec) stage_vector(r, i, tv[j]);
        }
      }
    };
    plain_row(0, ev);
    for (int r = 1; r < rows; ++r) {
      u32x4_t lv[NV], lg[NG];
      fetch_batch(r, 0, lv, lg);
      plain_row(r, lv);
    }
  }
  // the rest of the ring (see GEMV_RING_EARLY): in flight while the workgroup meets at the barrier
#pragma unroll
  for (int j = DE; j < D; ++j) {
    if (to_issue > 0) {
      DIHIP_GEMV_ISSUE(j);
      --to_issue;
    } else {
      DIHIP_GEMV_DUMMY(j);
    }
  }
  __syncthreads();
  DIHIP_GEMV_STAMP(3);

  // ---- main loop ---------------------------------------------------------------------------------
  float tot[MR], xacc[MR];
#pragma unroll
  for (int r = 0; r < MR; ++r) tot[r] = xacc[r] = 0.f;
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t g0 = zero4, g1 = zero4;
  uint32_t ex_mask = 0x000F000Fu, ex_magic = FT == DIHIP_BF16 ? 0x43004300u : 0x64006400u;
  asm volatile("" : "+v"(ex_mask), "+v"(ex_magic));  // opaque: keeps (x & mask) | magic a v_and_or_b32
  // consume cursor: half-unit cv, k-tile ckt, group countdown cgl.  A rows >= M read the zero block
  // (LDS byte address 0, no advance); the others walk their activation row.
  int cv = wn, ckt = k_lo, cgl = gcount;
  const bool arow_valid = ni < rows;
  const uint32_t xk_reset = arow_valid ? 256u + (uint32_t)(ni * a.RS + kb * 8 + k_lo * KTILE) * 2u : 0u;
  const uint32_t xk_step = arow_valid ? (uint32_t)KTILE * 2u : 0u;
  uint32_t xk = xk_reset;
  const float* xt = xsum_tab + k_lo * 16 + (MR == 1 ? 0 : kb * 4);

#define DIHIP_GEMV_CONSUME(SLOT)                                                                  \
  do {                                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < KSTEPS; ++ks) {                                       \
      const u32x4_t af_ = *reinterpret_cast<const u32x4_t*>(smem + xk + ks * 64);                 \
      const u32x4_t bf_ = EX::frag(wb[SLOT], ks, ex_mask, ex_magic);                              \
      if constexpr (QUANT && GPT) {                                                               \
        if (ks & 1) g1 = mfma16<FT>(af_, bf_, ks == 1 ? zero4 : g1);                              \
        else g0 = mfma16<FT>(af_, bf_, ks == 0 ? zero4 : g0);                                     \
      } else {                                                                                    \
        if (ks & 1) g1 = mfma16<FT>(af_, bf_, g1);                                                \
        else g0 = mfma16<FT>(af_, bf_, g0);                                                       \
      }                                                                                           \
    }                                                                                             \
    xk += xk_step;                                                                                \
    const bool tile_end_ = ++ckt == k_hi;                                                         \
    if constexpr (QUANT) {                                                                        \
      const float s_ = ft_bits_to_f32<FT>(sb[SLOT] & 0xFFFFu);                                    \
      const float nzp_ = -(ft_bits_to_f32<FT>(sb[SLOT] >> 16) + EX::OFFSET);                      \
      if constexpr (GPT) {                                                                        \
        if constexpr (MR == 1) {                                                                  \
          tot[0] = fmaf(s_, fmaf(nzp_, xt[0], KSTEPS > 1 ? g0[0] + g1[0] : g0[0]), tot[0]);       \
        } else {                                                                                  \
          const f32x4_t xv_ = *reinterpret_cast<const f32x4_t*>(xt);                              \
          _Pragma("unroll") for (int r = 0; r < MR; ++r)                                          \
            tot[r] = fmaf(s_, fmaf(nzp_, xv_[r], KSTEPS > 1 ? g0[r] + g1[r] : g0[r]), tot[r]);    \
        }                                                                                         \
        xt += 16;                                                                                 \
      } else {                                                                                    \
        if constexpr (MR == 1) {                                                                  \
          xacc[0] += xt[0];                                                                       \
        } else {                                                                                  \
          const f32x4_t xv_ = *reinterpret_cast<const f32x4_t*>(xt);                              \
          _Pragma("unroll") for (int r = 0; r < MR; ++r) xacc[r] += xv_[r];                       \
        }                                                                                         \
        xt += 16;                                                                                 \
        if (--cgl == 0 || tile_end_) { /* group end (wave-uniform) */                             \
          cgl = gcount;                                                                           \
          _Pragma("unroll") for (int r = 0; r < MR; ++r) {                                        \
            tot[r] = fmaf(s_, fmaf(nzp_, xacc[r], KSTEPS > 1 ? g0[r] + g1[r] : g0[r]), tot[r]);   \
            xacc[r] = 0.f;                                                                        \
          }                                                                                       \
          g0 = zero4;                                                                             \
          g1 = zero4;                                                                             \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
    if (tile_end_) {                                                                              \
      if constexpr (!QUANT) {                                                                     \
        _Pragma("unroll") for (int r = 0; r < MR; ++r) tot[r] = g0[r] + g1[r];                    \
        g0 = zero4;                                                                               \
        g1 = zero4;                                                                               \
      }                                                                                           \
      /* partial of (half-unit cv, k-slice wk): rows kb*4 + r, column ni */                       \
      float* dst_ = red + ((size_t)(cv * a.WK + wk) * rows) * 16 + ni;                            \
      _Pragma("unroll") for (int r = 0; r < MR; ++r) {                                            \
        const int m_ = kb * 4 + r;                                                                \
        if (m_ < rows) dst_[m_ * 16] = tot[r];                                                    \
        tot[r] = 0.f;                                                                             \
      }                                                                                           \
      ckt = k_lo;                                                                                 \
      cv += a.WN;                                                                                 \
      xk = xk_reset;                                                                              \
      xt = xsum_tab + k_lo * 16 + (MR == 1 ? 0 : kb * 4);                                         \
    }                                                                                             \
  } while (0)

  if (nk == 0) {
    // a k-slice without work (fewer groups than WK): its partials must read as zero
    for (int v = wn; v < nv; v += a.WN) {
      float* dst = red + ((size_t)(v * a.WK + wk) * rows) * 16;
      for (int i = lane; i < rows * 16; i += 64) dst[i] = 0.f;
    }
  }
  // Consume slot j -- everything older than the D-1 refills issued after it has landed -- and
  // refill it.  Steady state: every refill is a real chunk.
  // (The second-dispatched half of the workgroup streams ~13 % slower than the first -- the two waves of a SIMD are
  // arbitrated by age -- and finishes ~1.2 us later on the 72 MB launch.  Alternating s_setprio between the halves once
  // per ring revolution did not change that: 14.95 vs 14.96 us, profiles/r03b_*; removed.)
  int c = 0;
  while (to_issue >= D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      stream_wait<(D - 1) * LPC>();
      stream_landed(wb[j], sb[j]);
      DIHIP_GEMV_CONSUME(j);
      DIHIP_GEMV_ISSUE(j);
    }
    to_issue -= D;
    c += D;
  }

  // last real refills, then dummies
  for (; c < total; c += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (c + j >= total) break;  // (a break, not a guard: no path consumes slot j + 1 without slot j -- tools/audit_asm_loads.py)
      stream_wait<(D - 1) * LPC>();
      stream_landed(wb[j], sb[j]);
      DIHIP_GEMV_CONSUME(j);
      if (to_issue > 0) {
        DIHIP_GEMV_ISSUE(j);
        --to_issue;
      } else {
        DIHIP_GEMV_DUMMY(j);
      }
    }
  }
  // the ring registers die here: no load may still be in flight into them
  stream_wait<0>();
#pragma unroll
  for (int j = 0; j < D; ++j) asm volatile("" ::"v"(wb[j]), "v"(sb[j]));  // (alive until here: see DIHIP_GEMV_DUMMY)
#undef DIHIP_GEMV_ISSUE
#undef DIHIP_GEMV_DUMMY
#undef DIHIP_GEMV_CONSUME
  DIHIP_GEMV_STAMP(4);
  __syncthreads();
  DIHIP_GEMV_STAMP(5);

  // ---- combine the k-slices in fixed order + epilogue ----------------------------------------
  for (int e = tid; e < nu * rows * 16; e += GEMV_THREADS) {
    const int col = e & 15;
    const int m = MR == 1 ? 0 : (e >> 4) % rows;
    const int u = MR == 1 ? (e >> 4) : (e >> 4) / rows;
    const int n = (u0 + u * NB) * 16 + col;
    if (n >= a.N) continue;
    float v = 0.f, v2 = 0.f;
    const float* p = red + ((size_t)(u * DUAL) * a.WK * rows + m) * 16 + col;
    for (int s = 0; s < a.WK; ++s) v += p[(size_t)s * rows * 16];
    if constexpr (DUAL == 2) {
      const float* p2 = p + (size_t)a.WK * rows * 16;
      for (int s = 0; s < a.WK; ++s) v2 += p2[(size_t)s * rows * 16];
    }
    size_t yrow = (size_t)m * a.ldy;  // grouped slots: the row's slot
    if constexpr (SLOT && MR > 1) yrow = (size_t)a.slot_rows[(size_t)blockIdx.y * 4 + m] * a.ldy;
    if constexpr (EPI == EPI_STD) {
      v = __fmul_rn(a.alpha, v);
      if (a.bias) v = __fadd_rn(v, load_ft<FT>(a.bias, n));
      v = apply_act(v, a.act);
      if (a.residual) v = ft_round<FT>(v) + load_ft<FT>(a.residual, yrow + n);
      store_ft<FT>(p_y, yrow + n, v);
    } else if constexpr (EPI == EPI_SWIGLU) {
      if constexpr (HAND == 1) {
        // two neighbouring columns (lanes e, e + 1: same unit, same row) make one granule; every column tile is whole (N % 16 == 0)
        const uint32_t me = f32_to_ft_bits<FT>((v / (1.f + expf(-v))) * v2);
        const uint32_t nb = __shfl_down(me, 1, 64);
        if ((col & 1) == 0)
          __hip_atomic_store(a.hand_gran + (n >> 1), ((unsigned long long)hand_tag << 32) | (unsigned long long)(me | (nb << 16)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        store_ft<FT>(p_y, yrow + n, (v / (1.f + expf(-v))) * v2);
      }
    } else {
      const float base = a.h_res ? a.h_res[(size_t)m * a.N + n] : 0.f;
      const float hv = __fadd_rn(base, __fmul_rn(a.alpha, v));
      a.h_out[(size_t)m * a.N + n] = hv;
    }
  }
  if constexpr (HAND == 1) {
    // this workgroup's granules have left the CU (every storing wave drains), then its flag word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.hand_flags + bid, hand_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  DIHIP_GEMV_STAMP(6);
#undef DIHIP_GEMV_STAMP
}

template <int WBITS, int FT, int MR, int PRO, int EPI, int GPT, bool SLOT = false>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_stream_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gemv_stream_body<WBITS, FT, MR, PRO, EPI, GPT, SLOT>(a, (int)blockIdx.x, (int)gridDim.x, smem);
}

template <int WBITS, int FT, int MR, int PRO, int EPI, int GPT>
hipError_t launch_gemv_stream(const GemvArgs& a, int blocks, size_t lds_bytes, hipStream_t stream);
template <int WBITS, int FT, int EPI, int GPT, int MR>
hipError_t launch_gemv_slots(const GemvArgs& a, int blocks, size_t lds_bytes, hipStream_t stream);

#define DIHIP_DEFINE_GEMV_SLOT_LAUNCH(WBITS, FT, EPI, GPT, MR)                                                   \
  template <>                                                                                                    \
  hipError_t launch_gemv_slots<WBITS, FT, EPI, GPT, MR>(const GemvArgs& a, int blocks, size_t lds_bytes, hipStream_t s) { \
    auto kern = gemv_stream_kernel<WBITS, FT, MR, PRO_PLAIN, EPI, GPT, true>;                                    \
    hipLaunchKernelGGL(kern, dim3(blocks, a.nslots), dim3(GEMV_THREADS), lds_bytes, s, a);                       \
    return hipGetLastError();                                                                                    \
  }

#define DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, MR, PRO, EPI, GPT)                                    \
  template <>                                                                                     \
  hipError_t launch_gemv_stream<WBITS, FT, MR, PRO, EPI, GPT>(const GemvArgs& a, int blocks,      \
                                                              size_t lds_bytes, hipStream_t s) {  \
    auto kern = gemv_stream_kernel<WBITS, FT, MR, PRO, EPI, GPT>;                                 \
    if (lds_bytes > 64 * 1024) {                                                                  \
      static std::atomic<size_t> granted{0}; /* rank threads may race here: a second grant is harmless */ \
      if (lds_bytes > granted.load(std::memory_order_relaxed)) {                                  \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                           (int)lds_bytes);                                       \
        if (e != hipSuccess) return e;                                                            \
        granted.store(lds_bytes, std::memory_order_relaxed);                                      \
      }                                                                                           \
    }                                                                                             \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(GEMV_THREADS), lds_bytes, s, a);                  \
    return hipGetLastError();                                                                     \
  }

// the forms the decode step uses, for one (WBITS, FT, GPT)
#define DIHIP_DEFINE_GEMV_LAUNCH_SET(WBITS, FT, GPT)                  \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 1, PRO_PLAIN, EPI_STD, GPT)     \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 4, PRO_PLAIN, EPI_STD, GPT)     \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 1, PRO_RMSNORM, EPI_STD, GPT)   \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 4, PRO_RMSNORM, EPI_STD, GPT)   \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 1, PRO_RMSNORM, EPI_SWIGLU, GPT) \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 4, PRO_RMSNORM, EPI_SWIGLU, GPT) \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 1, PRO_PLAIN, EPI_SWIGLU, GPT)  \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 4, PRO_PLAIN, EPI_SWIGLU, GPT)  \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 1, PRO_PLAIN, EPI_ADDTO, GPT)   \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 4, PRO_PLAIN, EPI_ADDTO, GPT)   \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 1, PRO_RMSNORM, EPI_ADDTO, GPT) \
  DIHIP_DEFINE_GEMV_LAUNCH(WBITS, FT, 4, PRO_RMSNORM, EPI_ADDTO, GPT)

}  // namespace dihip
