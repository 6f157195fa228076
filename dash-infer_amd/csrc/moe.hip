// moe.hip -- mixture-of-experts decode path with weight-only quantised experts (BASELINE
// configs[4]: Qwen2-57B-A14B).  Semantics of the reference's MOE operator (csrc/core/operator/general/moe/moe_op.cpp:
// 338-460, inputs = hidden rows + router logits, weights = stacked gate_up / down projections, attributes num_experts,
// num_experts_per_tok):
//   float softmax over the experts (sum + 1e-12, csrc/core/kernel/cuda/softmax_low_reduce.cu:11-41), top-k of the
//   probabilities WITHOUT renormalisation (TopKKernelLauncher, moe_op.cpp:359-365), per (token, expert):
//   SiLU(x.Wgate) * (x.Wup) -> . Wdown  (UnaryGLU, csrc/core/kernel/cuda/unary.cu:122-132), and
//   out[t] = sum_k score[t,k] * y[t,k] accumulated in float in rank order (finalize_new_kernel, moe.cu:386-418).
// The reference has bf16 (MOE) and A8W8 experts only; here the experts are A16W8 / A16W4 weight-only (the quantised
// linear of section 1), which is what "MoE int8, expert GEMM" needs on a bandwidth-bound decode step.
//
// MI355X shape of the problem: a decode step touches top_k experts per token, each an M = 1 GEMV over weights nobody
// else reads -- the reference's reorder / pad / batched-GEMM machinery (moe_op.cpp:395-452) buys nothing.  One launch
// of the decode GEMV kernel per projection covers all (token, expert-rank) slots: gridDim.y = slots, the slot's
// expert index offsets the weight base (gemv_stream_kernel<..., SLOT = true>), SwiGLU is fused into the gate/up
// launch, and a small kernel applies the routing weights.  No host work per step: routing results stay on the device.
#include <algorithm>

#include "device_utils.h"
#include "gemm_lowp_kernel.hpp"  // mfma16

namespace dihip {

int run_gemv_slots(hipStream_t stream, int wbits, int epi, const void* x, int ldx, int x_div, const void* w0, const void* sz0,
                   const void* w1, const void* sz1, void* y, int N, int K, int group_size, const int* slot_expert, int nslots,
                   const int* group_rows, const int* group_nrows);
constexpr int MOE_EPI_STD = 0, MOE_EPI_SWIGLU = 1;  // EPI_STD / EPI_SWIGLU of gemm_lowp_kernel.hpp

struct ScoreIdx {
  float v;
  int i;
};
__device__ __forceinline__ ScoreIdx better(ScoreIdx a, ScoreIdx b) {  // larger value, lower index on ties
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

// the wave's best (value, index) in every lane: DPP moves inside the 16-lane rows, permlane swaps across them (the order of a
// total order's maximum does not matter); eight of these per token were 96 dependent ds_bpermute round trips with __shfl_xor
template <int CTRL>
__device__ __forceinline__ ScoreIdx dpp_pair(ScoreIdx a) {
  return ScoreIdx{dpp_mov_f32<CTRL>(a.v), __builtin_amdgcn_update_dpp(0, a.i, CTRL, 0xF, 0xF, true)};
}
__device__ __forceinline__ ScoreIdx wave_best(ScoreIdx a) {
  a = better(a, dpp_pair<0xB1>(a));
  a = better(a, dpp_pair<0x4E>(a));
  a = better(a, dpp_pair<0x141>(a));
  a = better(a, dpp_pair<0x140>(a));
  {
    const auto rv = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.v), __float_as_uint(a.v), false, false);
    const auto ri = __builtin_amdgcn_permlane32_swap((unsigned)a.i, (unsigned)a.i, false, false);
    a = better(ScoreIdx{__uint_as_float(rv[0]), (int)ri[0]}, ScoreIdx{__uint_as_float(rv[1]), (int)ri[1]});
  }
  {
    const auto rv = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.v), __float_as_uint(a.v), false, false);
    const auto ri = __builtin_amdgcn_permlane16_swap((unsigned)a.i, (unsigned)a.i, false, false);
    a = better(ScoreIdx{__uint_as_float(rv[0]), (int)ri[0]}, ScoreIdx{__uint_as_float(rv[1]), (int)ri[1]});
  }
  return a;
}

// one wave per token: softmax over E <= 256 router logits, then top_k picks in descending order
template <int FT>
__device__ __forceinline__ void route_token(float* __restrict__ scores, int* experts, const void* __restrict__ logits,
                                            int t, int lane, int E, int top_k, int ep_first, int ep_count) {
  float v[4];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = lane + j * 64;
    v[j] = e < E ? load_ft<FT>(logits, (size_t)t * E + e) : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = lane + j * 64 < E ? expf(v[j] - mx) : 0.f;
    sum += v[j];
  }
  sum = wave_sum(sum) + 1e-12f;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = lane + j * 64 < E ? v[j] / sum : -1.f;  // probabilities; -1 = not an expert
  for (int k = 0; k < top_k; ++k) {
    ScoreIdx best{-2.f, 0x7fffffff};
#pragma unroll
    for (int j = 0; j < 4; ++j) best = better(best, ScoreIdx{v[j], lane + j * 64});
    best = wave_best(best);
    if (lane == 0) {
      scores[(size_t)t * top_k + k] = best.v;
      // expert parallelism (moe_op.cpp:103-117: rank r owns experts [r * ep_num, (r + 1) * ep_num)): the index becomes the
      // position in this rank's stack, -1 for an expert that lives elsewhere (its slot is skipped, its term arrives by all-reduce)
      const int local = best.i - ep_first;
      experts[(size_t)t * top_k + k] = (local >= 0 && local < ep_count) ? local : -1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + j * 64 == best.i) v[j] = -1.f;  // taken
  }
}

template <int FT>
__global__ __launch_bounds__(64) void moe_route_kernel(float* __restrict__ scores, int* __restrict__ experts,
                                                        const void* __restrict__ logits, int E, int top_k, int ep_first,
                                                        int ep_count) {
  route_token<FT>(scores, experts, logits, blockIdx.x, threadIdx.x, E, top_k, ep_first, ep_count);
}

// out[t, :] = sum_k score[t, k] * y[t * top_k + k, :]  (float accumulation in rank order; skipped experts add nothing).
// grid (column chunks of 512, tokens): a thread owns two adjacent columns and issues the loads of up to 8 ranks together.
template <int FT>
__global__ __launch_bounds__(256) void moe_finalize_kernel(void* __restrict__ out, const void* __restrict__ y,
                                                           const float* __restrict__ scores, const int* __restrict__ experts,
                                                           int top_k, int cols) {
  const int t = blockIdx.y;
  const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (c >= cols) return;
  const bool pair = c + 1 < cols;
  float a0 = 0.f, a1 = 0.f;
  for (int k0 = 0; k0 < top_k; k0 += 8) {
    float v0[8], v1[8], sc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = min(k0 + j, top_k - 1);
      const size_t s = (size_t)t * top_k + k;
      const bool live = k0 + j < top_k && experts[s] >= 0;
      sc[j] = live ? scores[s] : 0.f;
      v0[j] = live ? load_ft<FT>(y, s * cols + c) : 0.f;
      v1[j] = live && pair ? load_ft<FT>(y, s * cols + c + 1) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 = a0 + sc[j] * v0[j];
      a1 = a1 + sc[j] * v1[j];
    }
  }
  store_ft<FT>(out, (size_t)t * cols + c, a0);
  if (pair) store_ft<FT>(out, (size_t)t * cols + c + 1, a1);
}

// Tail of the reference's MoE layer graph (python/pyhie/allspark/model/qwen_v20_moe.py:366-382) in one pass over the rows:
//   CalcExpert  c = shared_down * sigmoid_gate[t]   (calc_expert.cu:27-35; rounded to FT like the operator's output tensor)
//   expert_add + final_add:  h_out = h_res + moe_out + c   in f32 (the decoder's residual stream is f32; h_res == nullptr on
//   the ranks that do not carry the residual under tensor / expert parallelism: the sum over ranks happens in the all-reduce)
template <int FT>
__global__ __launch_bounds__(256) void moe_shared_combine_kernel(float* __restrict__ h_out, const float* __restrict__ h_res,
                                                                 const void* __restrict__ moe_out, const void* __restrict__ shared_out,
                                                                 const void* __restrict__ shared_gate, int cols) {
  const int t = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const size_t i = (size_t)t * cols + c;
  const float gate = load_ft<FT>(shared_gate, t);
  const float calc = ft_round<FT>(load_ft<FT>(shared_out, i) * gate);
  const float base = h_res ? h_res[i] : 0.f;
  h_out[i] = (base + load_ft<FT>(moe_out, i)) + calc;
}

// Slots that picked the same expert, gathered into groups of up to 4 (one workgroup; slots <= a few thousand): a group streams
// its expert ONCE for all its rows (gemv_stream_kernel<.., MR = 4, SLOT>), where the per-slot launch streams it once per token
// that picked it -- at 16 tokens x top-8 of 64 experts that is 128 expert reads for ~57 distinct experts.  The reference gets
// the same effect from its reorder / pad / batched-GEMM machinery (moe_op.cpp:395-452).  Deterministic: expert e's slots keep
// their order, groups are numbered by (expert, chunk).  Outputs for `nslots` group entries (unused ones: expert -1).
// All in LDS (round 3: the first form had every thread scan the slot list in global memory twice -- 21.8 us for 128 slots,
// profiles/r03u): slot s learns its rank among the earlier slots of its expert, the experts' group counts are scanned, and the
// slot writes itself into group  offset[expert] + rank / 4, row rank % 4.  Every thread of the block calls it (barriers inside).
constexpr int MOE_GROUP_LDS_SLOTS = 2048;
struct GroupLds {
  int exp[MOE_GROUP_LDS_SLOTS];
  int cnt[256], scan[256];
};
__device__ __forceinline__ void group_slots(int* __restrict__ group_expert, int* __restrict__ group_rows, int* __restrict__ group_nrows,
                                            const int* experts, int nslots, GroupLds& L) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int s = tid; s < nslots; s += nt) L.exp[s] = experts[s];
  if (tid < 256) L.cnt[tid] = 0;
  __syncthreads();
  for (int s = tid; s < nslots; s += nt)
    if (L.exp[s] >= 0) atomicAdd(&L.cnt[L.exp[s]], 1);
  __syncthreads();
  // exclusive scan of the experts' group counts (Hillis-Steele over 256 entries)
  const int mine = tid < 256 ? (L.cnt[tid] + 3) >> 2 : 0;
  if (tid < 256) L.scan[tid] = mine;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int add = (tid < 256 && tid >= d) ? L.scan[tid - d] : 0;
    __syncthreads();
    if (tid < 256) L.scan[tid] += add;
    __syncthreads();
  }
  const int total = L.scan[255];
  for (int s = tid; s < nslots; s += nt) {
    const int e = L.exp[s];
    if (e < 0) continue;
    int r = 0;
    for (int s2 = 0; s2 < s; ++s2) r += L.exp[s2] == e ? 1 : 0;  // rank among the expert's earlier slots (its order is kept)
    const int g = L.scan[e] - ((L.cnt[e] + 3) >> 2) + (r >> 2);
    group_rows[(size_t)g * 4 + (r & 3)] = s;
    if ((r & 3) == 0) {
      group_expert[g] = e;
      group_nrows[g] = min(4, L.cnt[e] - (r & ~3));
    }
  }
  for (int i = total + tid; i < nslots; i += nt) {  // entries past the last group
    group_expert[i] = -1;
    group_nrows[i] = 0;
  }
}

__global__ __launch_bounds__(256) void moe_group_kernel(int* __restrict__ group_expert, int* __restrict__ group_rows,
                                                        int* __restrict__ group_nrows, const int* __restrict__ experts, int nslots) {
  __shared__ GroupLds L;
  group_slots(group_expert, group_rows, group_nrows, experts, nslots, L);
}

// Routing and grouping of a decode batch in ONE launch (one workgroup of 16 waves: a wave routes tokens w, w + 16, ...; then the
// block groups the slots it has just written -- made visible by the fence + barrier; the loads below were not cached before)
template <int FT>
__global__ __launch_bounds__(1024) void moe_route_group_kernel(float* __restrict__ scores, int* experts,
                                                                const void* __restrict__ logits, int T, int E, int top_k, int ep_first,
                                                                int ep_count, int* __restrict__ group_expert, int* __restrict__ group_rows,
                                                                int* __restrict__ group_nrows) {
  __shared__ GroupLds L;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = wave; t < T; t += 16) route_token<FT>(scores, experts, logits, t, lane, E, top_k, ep_first, ep_count);
  // workgroup scope is enough (the readers are this workgroup: one CU, one L1, and these lines were not read before the stores);
  // a device-scope fence here writes the whole L2 back from every wave -- measured +13 us per layer
  __threadfence_block();
  __syncthreads();
  group_slots(group_expert, group_rows, group_nrows, experts, T * top_k, L);
}

// finalize-routing + the layer tail in one pass (moe_finalize_kernel's sum, rounded to FT like its output tensor, then
// moe_shared_combine_kernel's expression): bit-identical to the two launches
template <int FT>
__global__ __launch_bounds__(256) void moe_combine_kernel(float* __restrict__ h_out, const float* __restrict__ h_res,
                                                          const void* __restrict__ y, const float* __restrict__ scores,
                                                          const int* __restrict__ experts, const void* __restrict__ shared_out,
                                                          const void* __restrict__ shared_gate, int top_k, int cols) {
  const int t = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float a0 = 0.f;
  for (int k0 = 0; k0 < top_k; k0 += 8) {
    float v0[8], sc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = min(k0 + j, top_k - 1);
      const size_t s = (size_t)t * top_k + k;
      const bool live = k0 + j < top_k && experts[s] >= 0;
      sc[j] = live ? scores[s] : 0.f;
      v0[j] = live ? load_ft<FT>(y, s * cols + c) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a0 = a0 + sc[j] * v0[j];
  }
  const size_t i = (size_t)t * cols + c;
  const float gate = load_ft<FT>(shared_gate, t);
  const float calc = ft_round<FT>(load_ft<FT>(shared_out, i) * gate);
  const float base = h_res ? h_res[i] : 0.f;
  h_out[i] = (base + ft_round<FT>(a0)) + calc;
}

// The two unquantised skinny GEMMs of the MoE layer graph in ONE launch (qwen_v20_moe.py:330-338, 360-365): router logits
// = FT(xn . W_router) [T, E] and the shared expert's gate = FT(sigmoid(xn . w_gate)) [T, 1].  One workgroup of 16 waves per
// 16-column tile (E / 16 router tiles + the gate's tile); the waves split K (k-steps w, w + 16, ...: A fragment straight from
// the activation rows, B fragment = the packed chunk), meet in LDS and wave 0 sums the 16 partial tiles in fixed order.  The
// general kernel needs a split-K slab and a last-arriver reduction for such a shape: 2 x 9.0 us per layer (profiles/r03w).
template <int FT>
__global__ __launch_bounds__(1024) void moe_router_gate_kernel(void* __restrict__ logits, void* __restrict__ gate_out,
                                                                const void* __restrict__ xn, const u32x4_t* __restrict__ w_router,
                                                                const u32x4_t* __restrict__ w_gate, int T, int E, int K, int KT,
                                                                int router_tiles) {
  __shared__ f32x4_t part[16][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ni = lane & 15, kb = lane >> 4;
  const int tile = blockIdx.x;
  const bool is_gate = tile >= router_tiles;
  const u32x4_t* wp = (is_gate ? w_gate : w_router + (size_t)tile * KT * 64) + lane;
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int m0 = 0; m0 < T; m0 += 16) {
    const int row = min(m0 + ni, T - 1);  // rows past T re-read the last one (masked on store)
    const char* xr = reinterpret_cast<const char*>(xn) + ((size_t)row * K + kb * 8) * 2;
    f32x4_t acc = zero4;
    for (int ks = wave; ks < KT; ks += 16) {
      const u32x4_t af = *reinterpret_cast<const u32x4_t*>(xr + (size_t)ks * 64);
      const u32x4_t bf = wp[(size_t)ks * 64];
      acc = mfma16<FT>(af, bf, acc);
    }
    // compiler trap: the loop ends in the MFMA and hipcc puts the ds_write of its result straight behind the loop exit -- no
    // wait states for the matrix pipe's write-back across the block boundary (MFMA -> LDS read of the result is a software
    // hazard; MFMA -> VALU is interlocked): three of the four accumulator registers reached LDS without the last k-step.
    // Explicit wait states (24 >= the 19 a 16-pass MFMA needs) close it.
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
      f32x4_t v = part[0][lane];
#pragma unroll
      for (int w = 1; w < 16; ++w) {
        const f32x4_t t = part[w][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += t[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + kb * 4 + r;
        if (m >= T) continue;
        if (is_gate) {
          if (ni == 0) store_ft<FT>(gate_out, m, apply_act(v[r], DIHIP_ACT_SIGMOID));
        } else {
          const int n = tile * 16 + ni;
          if (n < E) store_ft<FT>(logits, (size_t)m * E + n, v[r]);
        }
      }
    }
    __syncthreads();
  }
}

// CalcExpert (csrc/core/kernel/cuda/calc_expert.cu:27-35): out[t, c] = in[t, c] * expert_weight[t]
template <int FT>
__global__ __launch_bounds__(256) void calc_expert_kernel(void* __restrict__ out, const void* __restrict__ in,
                                                          const void* __restrict__ expert_weight, int cols) {
  const int t = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const size_t i = (size_t)t * cols + c;
  store_ft<FT>(out, i, load_ft<FT>(in, i) * load_ft<FT>(expert_weight, t));
}

}  // namespace dihip

using namespace dihip;

extern "C" {

int dihip_moe_route(void* stream, const void* router_logits, int num_tokens, int num_experts, int top_k, float* scores,
                    int32_t* experts, int dtype) {
  return dihip_moe_route_ep(stream, router_logits, num_tokens, num_experts, top_k, scores, experts, dtype, 0, num_experts);
}

int dihip_moe_route_ep(void* stream, const void* router_logits, int num_tokens, int num_experts, int top_k, float* scores,
                       int32_t* experts, int dtype, int ep_first, int ep_count) {
  DIHIP_REQUIRE(ep_first >= 0 && ep_count > 0 && ep_first + ep_count <= num_experts, DIHIP_PARAM_ERROR,
                "moe_route: expert window [%d, %d) outside the %d experts", ep_first, ep_first + ep_count, num_experts);
  DIHIP_REQUIRE(num_tokens >= 0 && num_experts > 0 && num_experts <= 256 && top_k > 0 && top_k <= num_experts, DIHIP_PARAM_ERROR,
                "moe_route: need 0 < top_k <= num_experts <= 256 (moe_op.cpp:69-73)");
  DIHIP_REQUIRE(router_logits && scores && experts, DIHIP_PARAM_ERROR, "moe_route: null pointer");
  if (num_tokens == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DIHIP_BF16)
    hipLaunchKernelGGL(moe_route_kernel<DIHIP_BF16>, dim3(num_tokens), dim3(64), 0, s, scores, experts, router_logits, num_experts, top_k, ep_first, ep_count);
  else if (dtype == DIHIP_F16)
    hipLaunchKernelGGL(moe_route_kernel<DIHIP_F16>, dim3(num_tokens), dim3(64), 0, s, scores, experts, router_logits, num_experts, top_k, ep_first, ep_count);
  else if (dtype == DIHIP_F32)
    hipLaunchKernelGGL(moe_route_kernel<DIHIP_F32>, dim3(num_tokens), dim3(64), 0, s, scores, experts, router_logits, num_experts, top_k, ep_first, ep_count);
  else
    DIHIP_REQUIRE(false, DIHIP_PARAM_ERROR, "moe_route: unsupported dtype %d", dtype);
  return launch_status();
}

size_t dihip_moe_workspace_bytes(int num_tokens, int top_k, int hidden, int proj) {
  if (num_tokens <= 0 || top_k <= 0 || hidden <= 0 || proj <= 0) return 0;
  const size_t slots = (size_t)num_tokens * top_k;
  return (slots * proj * 2 + 255) / 256 * 256 + (slots * hidden * 2 + 255) / 256 * 256 + slots * 6 * sizeof(int) + 256;  // + group tables
}

namespace {
// the workspace of the block: [slots, proj] SiLU(gate) * up | [slots, hidden] expert outputs | group tables (expert [slots], rows
// [slots][4], nrows [slots])
struct MoeWs {
  char* act;
  char* ys;
  int *group_expert, *group_rows, *group_nrows;
};
MoeWs moe_ws_layout(void* ws, size_t slots, int hidden, int proj) {
  MoeWs w;
  w.act = reinterpret_cast<char*>(ws);
  w.ys = w.act + (slots * proj * 2 + 255) / 256 * 256;
  w.group_expert = reinterpret_cast<int*>(w.ys + (slots * hidden * 2 + 255) / 256 * 256);
  w.group_rows = w.group_expert + slots;
  w.group_nrows = w.group_rows + slots * 4;
  return w;
}
constexpr size_t MOE_GROUP_MAX_SLOTS = MOE_GROUP_LDS_SLOTS;  // the grouping pass is ONE workgroup with the slot list in LDS
}  // namespace

int dihip_moe_route_grouped(void* stream, const void* router_logits, int num_tokens, int num_experts, int top_k, float* scores,
                            int32_t* experts, int dtype, int ep_first, int ep_count, int hidden, int proj, void* ws, size_t ws_bytes) {
  DIHIP_REQUIRE(ep_first >= 0 && ep_count > 0 && ep_first + ep_count <= num_experts, DIHIP_PARAM_ERROR,
                "moe_route_grouped: expert window [%d, %d) outside the %d experts", ep_first, ep_first + ep_count, num_experts);
  DIHIP_REQUIRE(num_tokens > 1 && num_experts > 0 && num_experts <= 256 && top_k > 0 && top_k <= num_experts, DIHIP_PARAM_ERROR,
                "moe_route_grouped: need more than one token and 0 < top_k <= num_experts <= 256");
  const size_t slots = (size_t)num_tokens * top_k;
  DIHIP_REQUIRE(slots <= MOE_GROUP_MAX_SLOTS, DIHIP_EXCEED_LIMIT_ERROR,
                "moe_route_grouped: %zu slots: the one-launch form serves decode batches (<= %zu slots); use dihip_moe_route_ep", slots,
                MOE_GROUP_MAX_SLOTS);
  DIHIP_REQUIRE(router_logits && scores && experts && ws && ws_bytes >= dihip_moe_workspace_bytes(num_tokens, top_k, hidden, proj),
                DIHIP_PARAM_ERROR, "moe_route_grouped: null pointer or workspace too small");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const MoeWs w = moe_ws_layout(ws, slots, hidden, proj);
  if (dtype == DIHIP_BF16)
    hipLaunchKernelGGL(moe_route_group_kernel<DIHIP_BF16>, dim3(1), dim3(1024), 0, s, scores, experts, router_logits, num_tokens, num_experts,
                       top_k, ep_first, ep_count, w.group_expert, w.group_rows, w.group_nrows);
  else if (dtype == DIHIP_F16)
    hipLaunchKernelGGL(moe_route_group_kernel<DIHIP_F16>, dim3(1), dim3(1024), 0, s, scores, experts, router_logits, num_tokens, num_experts,
                       top_k, ep_first, ep_count, w.group_expert, w.group_rows, w.group_nrows);
  else
    DIHIP_REQUIRE(false, DIHIP_PARAM_ERROR, "moe_route_grouped: 16-bit router logits only (dtype %d)", dtype);
  return launch_status();
}

int dihip_moe_experts_ex(void* stream, int wbits, const void* x, const int32_t* experts, const float* scores,
                         const void* gate_packed, const void* gate_sz, const void* up_packed, const void* up_sz,
                         const void* down_packed, const void* down_sz, int num_tokens, int top_k, int hidden, int proj,
                         int group_size, void* out, void* ws, size_t ws_bytes, int dtype, int flags) {
  const bool pregrouped = flags & DIHIP_MOE_PREGROUPED, no_finalize = flags & DIHIP_MOE_NO_FINALIZE;
  DIHIP_REQUIRE(wbits == 4 || wbits == 8, DIHIP_PARAM_ERROR, "moe_experts: wbits must be 4 or 8");
  DIHIP_REQUIRE(dtype == DIHIP_BF16, DIHIP_PARAM_ERROR, "moe_experts: bf16 activations only");
  DIHIP_REQUIRE(num_tokens >= 0 && top_k > 0 && hidden > 0 && proj > 0, DIHIP_PARAM_ERROR, "moe_experts: bad shape");
  DIHIP_REQUIRE(x && experts && scores && gate_packed && gate_sz && up_packed && up_sz && down_packed && down_sz && (out || no_finalize),
                DIHIP_PARAM_ERROR, "moe_experts: null pointer");
  if (num_tokens == 0) return DIHIP_SUCCESS;
  const size_t slots = (size_t)num_tokens * top_k;
  DIHIP_REQUIRE(slots <= 65535, DIHIP_EXCEED_LIMIT_ERROR, "moe_experts: %zu (token, expert) slots exceed one launch (65535)", slots);
  DIHIP_REQUIRE(ws && ws_bytes >= dihip_moe_workspace_bytes(num_tokens, top_k, hidden, proj), DIHIP_MEMORY_ERROR,
                "moe_experts: workspace too small (%zu < %zu)", ws_bytes, dihip_moe_workspace_bytes(num_tokens, top_k, hidden, proj));
  DIHIP_REQUIRE(!pregrouped || (num_tokens > 1 && slots <= MOE_GROUP_MAX_SLOTS), DIHIP_PARAM_ERROR,
                "moe_experts: DIHIP_MOE_PREGROUPED needs the tables dihip_moe_route_grouped builds (1 < tokens, <= %zu slots)", MOE_GROUP_MAX_SLOTS);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const MoeWs w = moe_ws_layout(ws, slots, hidden, proj);
  // more than one token: slots of the same expert share one pass over its weights (moe_group_kernel); a single token's
  // top-k experts are distinct, the per-slot launch is the same work without the grouping launch
  const char* e = getenv("DIHIP_MOE_GROUP");  // =0: per-slot launches (diagnostics; read per call)
  const bool group_on = !(e && e[0] == '0');
  // (beyond MOE_GROUP_MAX_SLOTS the per-slot launches run instead of a long serial grouping launch)
  const bool grouped = pregrouped || (group_on && num_tokens > 1 && slots <= MOE_GROUP_MAX_SLOTS);
  const int* slot_expert = experts;
  const int *rows = nullptr, *nrows = nullptr;
  if (grouped) {
    if (!pregrouped)
      hipLaunchKernelGGL(moe_group_kernel, dim3(1), dim3(256), 0, s, w.group_expert, w.group_rows, w.group_nrows, experts, (int)slots);
    slot_expert = w.group_expert;
    rows = w.group_rows;
    nrows = w.group_nrows;
  }
  int st = run_gemv_slots(s, wbits, MOE_EPI_SWIGLU, x, hidden, top_k, gate_packed, gate_sz, up_packed, up_sz, w.act, proj, hidden,
                          group_size, slot_expert, (int)slots, rows, nrows);
  if (st) return st;
  st = run_gemv_slots(s, wbits, MOE_EPI_STD, w.act, proj, 1, down_packed, down_sz, nullptr, nullptr, w.ys, hidden, proj, group_size,
                      slot_expert, (int)slots, rows, nrows);
  if (st) return st;
  if (no_finalize) return launch_status();  // dihip_moe_combine sums the slots
  hipLaunchKernelGGL(moe_finalize_kernel<DIHIP_BF16>, dim3((hidden + 511) / 512, num_tokens), dim3(256), 0, s, out, w.ys, scores, experts,
                     top_k, hidden);
  return launch_status();
}

int dihip_moe_experts(void* stream, int wbits, const void* x, const int32_t* experts, const float* scores,
                      const void* gate_packed, const void* gate_sz, const void* up_packed, const void* up_sz,
                      const void* down_packed, const void* down_sz, int num_tokens, int top_k, int hidden, int proj,
                      int group_size, void* out, void* ws, size_t ws_bytes, int dtype) {
  return dihip_moe_experts_ex(stream, wbits, x, experts, scores, gate_packed, gate_sz, up_packed, up_sz, down_packed, down_sz, num_tokens,
                              top_k, hidden, proj, group_size, out, ws, ws_bytes, dtype, 0);
}

int dihip_moe_combine(void* stream, float* h_out, const float* h_res, const void* ws, const float* scores, const int32_t* experts,
                      const void* shared_out, const void* shared_gate, int num_tokens, int top_k, int hidden, int proj, int dtype) {
  DIHIP_REQUIRE(num_tokens >= 0 && top_k > 0 && hidden > 0 && proj > 0, DIHIP_PARAM_ERROR, "moe_combine: bad shape");
  DIHIP_REQUIRE(h_out && ws && scores && experts && shared_out && shared_gate, DIHIP_PARAM_ERROR, "moe_combine: null pointer");
  DIHIP_REQUIRE(dtype == DIHIP_BF16, DIHIP_PARAM_ERROR, "moe_combine: bf16 activations only (the expert GEMVs' type)");
  if (num_tokens == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const MoeWs w = moe_ws_layout(const_cast<void*>(ws), (size_t)num_tokens * top_k, hidden, proj);
  hipLaunchKernelGGL(moe_combine_kernel<DIHIP_BF16>, dim3((hidden + 255) / 256, num_tokens), dim3(256), 0, s, h_out, h_res, w.ys, scores,
                     experts, shared_out, shared_gate, top_k, hidden);
  return launch_status();
}

int dihip_moe_router_gate(void* stream, const void* xn, const void* w_router_packed, const void* w_gate_packed, void* router_logits,
                          void* shared_gate, int num_tokens, int num_experts, int hidden, int dtype) {
  DIHIP_REQUIRE(num_tokens >= 0 && num_experts > 0 && hidden > 0 && hidden % 32 == 0, DIHIP_PARAM_ERROR,
                "moe_router_gate: hidden must be a multiple of 32 (the packed k-step)");
  DIHIP_REQUIRE(xn && w_router_packed && w_gate_packed && router_logits && shared_gate, DIHIP_PARAM_ERROR, "moe_router_gate: null pointer");
  if (num_tokens == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rt = (num_experts + 15) / 16, KT = hidden / 32;
  const u32x4_t* wr = reinterpret_cast<const u32x4_t*>(w_router_packed);
  const u32x4_t* wg = reinterpret_cast<const u32x4_t*>(w_gate_packed);
  if (dtype == DIHIP_BF16)
    hipLaunchKernelGGL(moe_router_gate_kernel<DIHIP_BF16>, dim3(rt + 1), dim3(1024), 0, s, router_logits, shared_gate, xn, wr, wg, num_tokens,
                       num_experts, hidden, KT, rt);
  else if (dtype == DIHIP_F16)
    hipLaunchKernelGGL(moe_router_gate_kernel<DIHIP_F16>, dim3(rt + 1), dim3(1024), 0, s, router_logits, shared_gate, xn, wr, wg, num_tokens,
                       num_experts, hidden, KT, rt);
  else
    DIHIP_REQUIRE(false, DIHIP_PARAM_ERROR, "moe_router_gate: 16-bit activations only (dtype %d)", dtype);
  return launch_status();
}

int dihip_moe_shared_combine(void* stream, float* h_out, const float* h_res, const void* moe_out, const void* shared_out,
                             const void* shared_gate, int num_tokens, int hidden, int dtype) {
  DIHIP_REQUIRE(num_tokens >= 0 && hidden > 0, DIHIP_PARAM_ERROR, "moe_shared_combine: bad shape");
  DIHIP_REQUIRE(h_out && moe_out && shared_out && shared_gate, DIHIP_PARAM_ERROR, "moe_shared_combine: null pointer");
  if (num_tokens == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((hidden + 255) / 256, num_tokens);
  if (dtype == DIHIP_BF16)
    hipLaunchKernelGGL(moe_shared_combine_kernel<DIHIP_BF16>, grid, dim3(256), 0, s, h_out, h_res, moe_out, shared_out, shared_gate, hidden);
  else if (dtype == DIHIP_F16)
    hipLaunchKernelGGL(moe_shared_combine_kernel<DIHIP_F16>, grid, dim3(256), 0, s, h_out, h_res, moe_out, shared_out, shared_gate, hidden);
  else
    DIHIP_REQUIRE(false, DIHIP_PARAM_ERROR, "moe_shared_combine: 16-bit activations only (dtype %d)", dtype);
  return launch_status();
}

int dihip_calc_expert(void* stream, void* out, const void* in, const void* expert_weight, int num_tokens, int hidden, int dtype) {
  DIHIP_REQUIRE(num_tokens >= 0 && hidden > 0 && out && in && expert_weight, DIHIP_PARAM_ERROR, "calc_expert: bad argument");
  if (num_tokens == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((hidden + 255) / 256, num_tokens);
  if (dtype == DIHIP_BF16)
    hipLaunchKernelGGL(calc_expert_kernel<DIHIP_BF16>, grid, dim3(256), 0, s, out, in, expert_weight, hidden);
  else if (dtype == DIHIP_F16)
    hipLaunchKernelGGL(calc_expert_kernel<DIHIP_F16>, grid, dim3(256), 0, s, out, in, expert_weight, hidden);
  else if (dtype == DIHIP_F32)
    hipLaunchKernelGGL(calc_expert_kernel<DIHIP_F32>, grid, dim3(256), 0, s, out, in, expert_weight, hidden);
  else
    DIHIP_REQUIRE(false, DIHIP_PARAM_ERROR, "calc_expert: unsupported dtype %d", dtype);
  return launch_status();
}

}  // extern "C"
