// prefill_attn.hip -- causal GQA prefill attention on the gfx950 matrix cores
// (include/dashinfer_hip.h section 4).
//
// Replaces xformer_prefill_attention (csrc/core/kernel/cuda/xformer_mha/xformer_mha.h:26-41; CUTLASS
// fMHA with kBlockQuery = 32, kBlockKey = 128, xformer_mha.dispatch.cu:37-41) with a
// flash-style single pass written for wave64 MFMA:
//   * workgroup = 4 waves = 64 query rows of one head; each wave owns 16 rows and keeps its Q
//     fragments (A operand of v_mfma_f32_16x16x32, 16 VGPRs for H = 128) and the 16 x 128 f32
//     output accumulator (8 C tiles) in registers for the whole pass;
//   * K and V tiles of 32 keys are staged once per workgroup in LDS: K row-major (its rows are
//     the B fragments of Q.K^T as stored), V transposed on the way in (so that the B fragments
//     of P.V are contiguous 16-byte reads);
//   * scores stay in f32: S = alpha * Q.K^T -> causal mask -> online softmax (row max / sum over
//     the 16 key lanes with DPP row rotations) -> P rounded to FT -> through a 1 KiB per-wave LDS
//     patch from the C layout into the A layout -> P.V;
//   * key tiles entirely above the causal diagonal are skipped.
// Numerics follow the reference's CPU check (tests/cpp/kernel/cuda/kernel_mhaprefill_test.cpp:
// 119-320): f32 softmax, FT inputs/outputs; P is rounded to FT before P.V as the tensor-core
// kernels of the reference do.
#include <algorithm>

#include "gemm_lowp_kernel.hpp"  // mfma16<FT>

namespace dihip {

constexpr int PF_THREADS = 256;
constexpr int PF_QROWS = 64;   // query rows per workgroup (16 per wave)
constexpr int PF_KEYS = 32;    // keys per tile
constexpr int PF_KPITCH = 136; // K tile row pitch in elements (128 + 8: conflict-free b128 reads)
constexpr int PF_VPITCH = 40;  // V^T tile row pitch in elements (32 + 8)

struct PrefillArgs {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  int seq_q, seq_k, q_stride, kv_stride, n_heads, n_groups, causal;
  float alpha;
};

__device__ __forceinline__ float row16_max_pf(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true)));
  return v;
}
__device__ __forceinline__ float row16_sum_pf(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true));
  return v;
}

template <int FT>
__global__ __launch_bounds__(PF_THREADS) void prefill_attn_kernel(const PrefillArgs a) {
  constexpr int H = 128;
  __shared__ __attribute__((aligned(16))) uint16_t ks[PF_KEYS * PF_KPITCH];   // K tile [key][dim]
  __shared__ __attribute__((aligned(16))) uint16_t vt[H * PF_VPITCH];         // V tile transposed [dim][key]
  __shared__ __attribute__((aligned(16))) uint16_t ps[4 * 16 * PF_VPITCH];    // per-wave P patch [q row][key]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ni = lane & 15, kb = lane >> 4;
  const int head = blockIdx.y, kvh = head / (a.n_heads / a.n_groups);
  const int q0 = blockIdx.x * PF_QROWS + wave * 16;  // first query row of this wave
  const int shift = a.seq_k - a.seq_q;               // query i sees keys j <= i + shift

  // Q fragments: lane (kb, ni) holds Q[q0 + ni][ks*32 + kb*8 .. +8]
  u32x4_t qf[4];
  {
    const int qr = min(q0 + ni, a.seq_q - 1);
    const uint16_t* qp = reinterpret_cast<const uint16_t*>(a.q) + (size_t)qr * a.q_stride + (size_t)head * H + kb * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const u32x4_t*>(qp + s * 32);
  }
  f32x4_t oacc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) oacc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float mrow[4], lrow[4];  // running max / sum of query rows kb*4 + r (replicated over the 16 key lanes)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    mrow[r] = -INFINITY;
    lrow[r] = 0.f;
  }

  // keys needed by this workgroup: up to the diagonal of its last query row
  const int wg_last_q = min(a.seq_q, (int)(blockIdx.x + 1) * PF_QROWS) - 1;
  const int k_end = a.causal ? min(a.seq_k, wg_last_q + shift + 1) : a.seq_k;
  uint16_t* pw = ps + wave * 16 * PF_VPITCH;

  for (int k0 = 0; k0 < k_end; k0 += PF_KEYS) {
    __syncthreads();  // previous tile fully consumed
    // ---- stage K (row-major) and V (transposed): 32 keys x 128 dims each, 2 x 16 B per thread ----
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + it * PF_THREADS;   // 0..511: key = c / 16, dim chunk = c % 16
      const int key = c >> 4, dc = c & 15;
      const int kr = min(k0 + key, a.seq_k - 1);
      const size_t off = (size_t)kr * a.kv_stride + (size_t)kvh * H + dc * 8;
      const u32x4_t kv4 = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(a.k) + off);
      const u32x4_t vv4 = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const uint16_t*>(a.v) + off);
      *reinterpret_cast<u32x4_t*>(ks + key * PF_KPITCH + dc * 8) = kv4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vt[(dc * 8 + 2 * j) * PF_VPITCH + key] = (uint16_t)(vv4[j] & 0xFFFFu);
        vt[(dc * 8 + 2 * j + 1) * PF_VPITCH + key] = (uint16_t)(vv4[j] >> 16);
      }
    }
    __syncthreads();
    const bool wave_live = !a.causal || k0 <= q0 + 15 + shift;  // some key of the tile is visible to this wave
    if (wave_live && q0 < a.seq_q) {
      // ---- S = alpha * Q.K^T for 2 key tiles of 16 ----
      f32x4_t sacc[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        sacc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const uint16_t* kp = ks + (nt * 16 + ni) * PF_KPITCH + kb * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const u32x4_t bf = *reinterpret_cast<const u32x4_t*>(kp + s * 32);
          sacc[nt] = mfma16<FT>(qf[s], bf, sacc[nt]);
        }
      }
      // ---- mask + online softmax: lane holds S[q = q0 + kb*4 + r][key = k0 + nt*16 + ni] ----
      float p[2][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + kb * 4 + r;
        float mx = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int key = k0 + nt * 16 + ni;
          const bool vis = key < a.seq_k && (!a.causal || key <= qi + shift);
          const float sv = vis ? sacc[nt][r] * a.alpha : -INFINITY;
          p[nt][r] = sv;
          mx = fmaxf(mx, sv);
        }
        mx = row16_max_pf(mx);
        const float mn = fmaxf(mrow[r], mx);
        const float corr = mrow[r] == -INFINITY ? 0.f : __expf(mrow[r] - mn);
        float psum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float e = p[nt][r] == -INFINITY ? 0.f : __expf(p[nt][r] - mn);
          const float er = ft_round<FT>(e);  // P is fed to the matrix core in FT
          p[nt][r] = er;
          psum += er;
        }
        psum = row16_sum_pf(psum);
        lrow[r] = lrow[r] * corr + psum;
        mrow[r] = mn;
#pragma unroll
        for (int t = 0; t < 8; ++t) oacc[t][r] *= corr;
      }
      // ---- P: C layout -> A layout through the wave's LDS patch [q row][32 keys] ----
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(kb * 4 + r) * PF_VPITCH + nt * 16 + ni] = (uint16_t)f32_to_ft_bits<FT>(p[nt][r]);
      // same wave writes and reads: LDS ops of one wave complete in order
      const u32x4_t pf = *reinterpret_cast<const u32x4_t*>(pw + ni * PF_VPITCH + kb * 8);
      // ---- O += P.V: B fragment of dim tile t = V^T[t*16 + ni][kb*8 .. +8] ----
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const u32x4_t vf = *reinterpret_cast<const u32x4_t*>(vt + (t * 16 + ni) * PF_VPITCH + kb * 8);
        oacc[t] = mfma16<FT>(pf, vf, oacc[t]);
      }
    }
  }
  // ---- normalise and store: lane holds O[q0 + kb*4 + r][t*16 + ni] ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + kb * 4 + r;
    if (qi < a.seq_q) {
      const float inv = lrow[r] > 0.f ? 1.f / lrow[r] : 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        store_ft<FT>(a.out, (size_t)qi * a.n_heads * H + (size_t)head * H + t * 16 + ni, oacc[t][r] * inv);
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
  if (seq_q == 0) return DIHIP_SUCCESS;
  DIHIP_REQUIRE(out && q && k && v && seq_k > 0, DIHIP_PARAM_ERROR, "prefill_attn: null pointer / empty keys");
  DIHIP_REQUIRE(q_stride % 8 == 0 && kv_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(k) & 15) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0,
                DIHIP_PARAM_ERROR, "prefill_attn: rows must be 16-byte aligned");
  PrefillArgs a{out, q, k, v, seq_q, seq_k, q_stride, kv_stride, n_heads, n_groups, causal, alpha};
  const dim3 grid((seq_q + PF_QROWS - 1) / PF_QROWS, n_heads);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DIHIP_BF16) hipLaunchKernelGGL(prefill_attn_kernel<DIHIP_BF16>, grid, dim3(PF_THREADS), 0, s, a);
  else hipLaunchKernelGGL(prefill_attn_kernel<DIHIP_F16>, grid, dim3(PF_THREADS), 0, s, a);
  return launch_status();
}
