// decode_mlp_block.hip -- the feed-forward half of a batch-1 decode layer in ONE launch (include/dashinfer_hip.h section 3f):
//     RMSNorm + gate / up GEMV + SwiGLU   ->   down GEMV + residual
// i.e. LayerNormNoBeta -> Gemm[A16W4](gate, SILU) || Gemm[A16W4](up) -> Binary MUL -> Gemm[A16W4](down) -> Binary ADD of the reference
// graph (python/pyhie/allspark/model/qwen_v15.py:300-388).  With dihip_decode_attn_block a decode layer is TWO launches.
//
// Both phases are the bodies of the stand-alone decode GEMV (gemv_stream_kernel.hpp: same workgroup count, same K split, same
// sums -- bit-identical results), run one after the other by the same resident workgroups:
//   phase 1  gemv_stream_body<PRO_RMSNORM, EPI_SWIGLU, HAND 1>: the SwiGLU outputs leave as 8-byte granules {two bf16, tag} + one flag
//            word per workgroup once its granules have drained;
//   phase 2  gemv_stream_body<PRO_PLAIN, EPI_ADDTO, HAND 2>: the WHOLE weight ring of the down projection is requested first -- the
//            weights do not depend on the activations: 64 KB per workgroup stream while the slower producers finish -- then the
//            producers' flags are awaited, the 18944-element row is swept into LDS and the stream goes on.
// What the single launch removes: the first launch's tail, the boundary, and the second launch's launch -> first byte, during all of
// which HBM idles in the chain (profiles/r05_*).  Hand-off as in decode_attn_block.hip: agent-scope granules, the launch's own
// epoch as tag, every wait bounded (error word), all workgroups resident (<= one per CU, checked on the host).
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "gemv_stream_kernel.hpp"

namespace dihip {

bool gemv_plan_args(int wbits, int N, int K, int group_size, bool dual, GemvArgs* g, int* blocks, size_t* lds_bytes);

struct MlpBlockArgs {
  GemvArgs gu;    // RMSNorm + gate / up + SwiGLU
  GemvArgs down;  // down projection + residual
  unsigned* state;  // [0] epoch, [1] error, [3] arrival counter of the epoch hand-over (zero between launches)
  int nb_gu, nb_down;
};

__global__ __launch_bounds__(GEMV_THREADS) void decode_mlp_block_kernel(const MlpBlockArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bid = (int)blockIdx.x;
  unsigned tag = p.state[0] + 1u;
  if (tag == 0u) tag = 1u;
  // (the argument blocks are read where they lie, in the kernel-argument segment: the ring loads take wave-uniform bases in SGPRs)
  if (bid < p.nb_gu) gemv_stream_body<4, DIHIP_BF16, 1, PRO_RMSNORM, EPI_SWIGLU, 1, false, 1>(p.gu, bid, p.nb_gu, smem, tag);
  if (bid < p.nb_down) {
    __syncthreads();  // (the staging area is re-used)
    gemv_stream_body<4, DIHIP_BF16, 1, PRO_PLAIN, EPI_ADDTO, 1, false, 2>(p.down, bid, p.nb_down, smem, tag);
  }
  // The next launch's epoch is written by the LAST workgroup to finish (an arrival counter in state[3], left at zero): every workgroup
  // has read the old epoch by then -- its own arrival comes after.  (Round 5 let workgroup 0 write it once ITS waits were over, which
  // covers every PRODUCER only: with more down-projection blocks than gate / up blocks -- Qwen2-7B: 251 against 237 -- a consumer-only
  // workgroup could still be about to read the word.  ADVICE r5.)
  if (threadIdx.x == 0) {
    const unsigned total = (unsigned)max(p.nb_gu, p.nb_down);
    const unsigned t = __hip_atomic_fetch_add(p.state + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == total - 1u) {
      __hip_atomic_store(p.state + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.state + 0, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static bool mlp_block_enabled() {
  static const bool on = !env_off("DIHIP_MLP_BLOCK");  // =0: "not supported" (callers keep the two launches; A/B)
  return on;
}

struct MbLayout {
  size_t flags, gran, total;
};
static MbLayout mb_layout(int inter) {
  MbLayout l;
  l.flags = 64;
  l.gran = l.flags + 1024 * sizeof(unsigned);  // <= 1024 producer workgroups
  l.total = l.gran + (size_t)inter / 2 * 8;
  return l;
}

}  // namespace dihip

using namespace dihip;

extern "C" {

int dihip_decode_mlp_block_supported(int wbits, int group_size, int hidden, int inter, int dtype, int batch) {
  if (!mlp_block_enabled() || batch != 1 || wbits != 4 || dtype != DIHIP_BF16 || hidden <= 0 || inter <= 0 || inter % 16 || hidden % 16) return 0;
  GemvArgs g{}, d{};
  int bg, bd;
  size_t lg, ld;
  if (!gemv_plan_args(4, inter, hidden, group_size, true, &g, &bg, &lg) || g.ktpg != 1) return 0;
  if (!gemv_plan_args(4, hidden, inter, group_size, false, &d, &bd, &ld) || d.ktpg != 1) return 0;
  const int ncu = cached_num_cus();
  // every workgroup resident at once, one per CU (the launch has max(bg, bd) workgroups); the flag poll covers 4 * 64 producers
  return ncu > 0 && bg <= ncu && bd <= ncu && bg <= 256 && std::max(lg, ld) <= 150 * 1024 ? 1 : 0;
}

size_t dihip_decode_mlp_block_sync_bytes(int inter) { return inter > 0 ? mb_layout(inter).total : 0; }

int dihip_decode_mlp_block(void* stream, int wbits, const float* h_in, const float* h_res, float* h_out, const void* gamma, float eps,
                           const void* gate_w, const void* gate_sz, const void* up_w, const void* up_sz, const void* down_w,
                           const void* down_sz, int hidden, int inter, int group_size, int dtype, void* sync, size_t sync_bytes) {
  DIHIP_REQUIRE(h_in && h_out && gamma && gate_w && gate_sz && up_w && up_sz && down_w && down_sz && sync, DIHIP_PARAM_ERROR,
                "decode_mlp_block: null pointer");
  DIHIP_REQUIRE(dihip_decode_mlp_block_supported(wbits, group_size, hidden, inter, dtype, 1), DIHIP_PARAM_ERROR,
                "decode_mlp_block: configuration not covered (batch 1, bf16, int4 g128, decode-GEMV shapes); see _supported");
  DIHIP_REQUIRE(reinterpret_cast<uintptr_t>(h_in) % 16 == 0 && reinterpret_cast<uintptr_t>(gamma) % 16 == 0, DIHIP_PARAM_ERROR,
                "decode_mlp_block: the hidden row and gamma must be 16-byte aligned");
  const MbLayout lay = mb_layout(inter);
  DIHIP_REQUIRE(sync_bytes >= lay.total && reinterpret_cast<uintptr_t>(sync) % 16 == 0, DIHIP_MEMORY_ERROR,
                "decode_mlp_block: sync buffer too small (%zu < %zu)", sync_bytes, lay.total);
  MlpBlockArgs p{};
  size_t lg, ld;
  gemv_plan_args(4, inter, hidden, group_size, true, &p.gu, &p.nb_gu, &lg);
  gemv_plan_args(4, hidden, inter, group_size, false, &p.down, &p.nb_down, &ld);
  char* sb = reinterpret_cast<char*>(sync);
  static const unsigned spin_limit = (unsigned)std::max(1024, env_int("DIHIP_ATTN_BLOCK_SPINS", 1 << 18));
  for (GemvArgs* g : {&p.gu, &p.down}) {
    g->hand_gran = reinterpret_cast<unsigned long long*>(sb + lay.gran);
    g->hand_flags = reinterpret_cast<unsigned*>(sb + lay.flags);
    g->hand_err = reinterpret_cast<unsigned*>(sb) + 1;
    g->hand_spin_limit = spin_limit;
    g->hand_nproducers = p.nb_gu;
  }
  p.state = reinterpret_cast<unsigned*>(sb);
  p.gu.w0 = reinterpret_cast<const u32x4_t*>(gate_w);
  p.gu.w1 = reinterpret_cast<const u32x4_t*>(up_w);
  p.gu.sz0 = reinterpret_cast<const uint32_t*>(gate_sz);
  p.gu.sz1 = reinterpret_cast<const uint32_t*>(up_sz);
  p.gu.x = h_in;
  p.gu.gamma = gamma;
  p.gu.eps = eps;
  p.down.w0 = reinterpret_cast<const u32x4_t*>(down_w);
  p.down.sz0 = reinterpret_cast<const uint32_t*>(down_sz);
  p.down.h_res = h_res;
  p.down.h_out = h_out;
  // diagnostics (`make trace` build + dihip_debug_set_trace): [phase][workgroup][wave][8] wall-clock stamps of the two bodies
  {
    const size_t per = (size_t)std::max(p.nb_gu, p.nb_down) * GEMV_WAVES * 8;
    unsigned long long* tr = debug_trace_buffer(2 * per * sizeof(unsigned long long));
    p.gu.trace = tr;
    p.down.trace = tr ? tr + per : nullptr;
  }
  const size_t lds = std::max(lg, ld);
  auto kern = decode_mlp_block_kernel;
  if (lds > 64 * 1024) {
    static std::atomic<size_t> granted{0};
    if (lds > granted.load(std::memory_order_relaxed)) {
      DIHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                      DIHIP_RUNTIME_ERROR);
      granted.store(lds, std::memory_order_relaxed);
    }
  }
  hipLaunchKernelGGL(kern, dim3(std::max(p.nb_gu, p.nb_down)), dim3(GEMV_THREADS), lds, reinterpret_cast<hipStream_t>(stream), p);
  return launch_status();
}

}  // extern "C"
