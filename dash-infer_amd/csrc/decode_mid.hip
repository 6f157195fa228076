// decode_mid.hip -- two consecutive GEMVs of a decode layer in ONE launch (include/dashinfer_hip.h section 3e):
//   producer  h_out = h_res + attn . Wo                                   (the o-projection with its residual, EPI_ADDTO)
//   consumer  act   = SiLU(norm(h_out) . Wgate) * (norm(h_out) . Wup)      (RMSNorm + gate / up GEMV + SwiGLU)
// In the launch chain the second kernel starts from nothing behind a boundary: ~1.9 us until its weight ring is requested,
// another round trip until the first weights arrive.  None of that depends on the hidden row.  Here the consumer's
// workgroups are part of the producer's launch: they ramp and fill their weight ring (64 KB per CU in flight) while the
// producer streams, wait on completion counters the producer workgroups bump once their write-through stores of the row
// have been acknowledged, read the row past the L2 and carry on with the RMSNorm prologue -- the kernel boundary and the
// consumer's ramp leave the critical path.  Bit-identical to the two calls it replaces (same kernels' bodies, same plans).
//
// Co-residency: the producer workgroups have the low block indices (dispatched first) and never wait; the consumers only
// wait.  Both are 8-wave workgroups of < 100 VGPRs and together < 64 KB of LDS: two fit a CU, and the entry refuses
// shapes whose workgroups exceed 2 x CUs.  A wait that sees no progress for seconds traps.
#include <algorithm>
#include <string>

#include "gemv_stream_kernel.hpp"

namespace dihip {

bool gemv_mid_plan(int wbits, const void* x, const void* wo, const void* szo, const float* h_res, float* h_out, int No, int Ko,
                   const void* gamma, float eps, const void* wg, const void* szg, const void* wu, const void* szu, void* act,
                   int Ni, int group_size, GemvArgs* go, int* blocks_o, size_t* lds_o, GemvArgs* gg, int* blocks_g, size_t* lds_g,
                   int* gpt);

constexpr int MID_REPLICAS = 8;
constexpr int MID_SYNC_STRIDE = 32;  // one 128-byte line per counter

template <int WBITS, int GPT>
__global__ __launch_bounds__(GEMV_THREADS, 2) void decode_mid_kernel(const GemvArgs go, const GemvArgs gg, const int blocks_o,
                                                                  const int blocks_g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bid = (int)blockIdx.x;
  if (bid < blocks_o)
    gemv_stream_body<WBITS, DIHIP_BF16, 1, PRO_PLAIN, EPI_ADDTO, GPT, false, GEMV_SYNC_HPUB>(go, bid, blocks_o, smem);
  else
    gemv_stream_body<WBITS, DIHIP_BF16, 1, PRO_RMSNORM, EPI_SWIGLU, GPT, false, GEMV_SYNC_HWAIT>(gg, bid - blocks_o, blocks_g, smem);
}

// Sequential form (DIHIP_MID_MODE=seq): max(blocks_o, blocks_g) workgroups, one per CU; a workgroup runs its share of the
// producer, requests the consumer's weight ring, waits for all producers and runs its share of the consumer -- no second
// streaming workgroup on the CU, the hand-over wait overlaps the ring's first round trip.
template <int WBITS, int GPT>
__global__ __launch_bounds__(GEMV_THREADS, 2) void decode_mid_seq_kernel(const GemvArgs go, const GemvArgs gg, const int blocks_o,
                                                                      const int blocks_g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bid = (int)blockIdx.x;
  if (bid < blocks_o) {
    gemv_stream_body<WBITS, DIHIP_BF16, 1, PRO_PLAIN, EPI_ADDTO, GPT, false, GEMV_SYNC_HPUB>(go, bid, blocks_o, smem);
    __syncthreads();  // the LDS carve-up changes hands
  }
  if (bid < blocks_g)
    gemv_stream_body<WBITS, DIHIP_BF16, 1, PRO_RMSNORM, EPI_SWIGLU, GPT, false, GEMV_SYNC_HWAIT>(gg, bid, blocks_g, smem);
}

// The two bodies as kernels of their own.  Never launched: they exist so that tools/audit_asm_loads.py can walk each body's
// control flow by itself -- in the fused kernel the compiler joins the two bodies through a scalar flag, which a
// path-insensitive walk cannot follow (tests/test_asm_audit.py skips decode_mid_kernel and audits these).
template <int WBITS, int GPT>
__global__ __launch_bounds__(GEMV_THREADS, 2) void decode_mid_producer_audit_kernel(const GemvArgs go) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gemv_stream_body<WBITS, DIHIP_BF16, 1, PRO_PLAIN, EPI_ADDTO, GPT, false, GEMV_SYNC_HPUB>(go, (int)blockIdx.x, (int)gridDim.x, smem);
}
template <int WBITS, int GPT>
__global__ __launch_bounds__(GEMV_THREADS, 2) void decode_mid_consumer_audit_kernel(const GemvArgs gg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gemv_stream_body<WBITS, DIHIP_BF16, 1, PRO_RMSNORM, EPI_SWIGLU, GPT, false, GEMV_SYNC_HWAIT>(gg, (int)blockIdx.x, (int)gridDim.x, smem);
}
#define MID_AUDIT(W, G)                                                      \
  template __global__ void decode_mid_producer_audit_kernel<W, G>(GemvArgs); \
  template __global__ void decode_mid_consumer_audit_kernel<W, G>(GemvArgs);
MID_AUDIT(4, 1) MID_AUDIT(4, 0) MID_AUDIT(8, 1) MID_AUDIT(8, 0)
#undef MID_AUDIT

template <int WBITS, int GPT>
static hipError_t launch_mid(const GemvArgs& go, const GemvArgs& gg, int blocks_o, int blocks_g, size_t lds, hipStream_t s) {
  static int seq = -1;  // DIHIP_MID_MODE=seq: the sequential form (one workgroup per CU runs both)
  if (seq < 0) {
    const char* e = getenv("DIHIP_MID_MODE");
    seq = (e && std::string(e) == "seq") ? 1 : 0;
  }
  if (seq) {
    auto kern = decode_mid_seq_kernel<WBITS, GPT>;
    hipLaunchKernelGGL(kern, dim3(std::max(blocks_o, blocks_g)), dim3(GEMV_THREADS), lds, s, go, gg, blocks_o, blocks_g);
  } else {
    auto kern = decode_mid_kernel<WBITS, GPT>;
    hipLaunchKernelGGL(kern, dim3(blocks_o + blocks_g), dim3(GEMV_THREADS), lds, s, go, gg, blocks_o, blocks_g);
  }
  return hipGetLastError();
}

}  // namespace dihip

using namespace dihip;

extern "C" {

size_t dihip_decode_mid_sync_bytes(void) { return (size_t)(MID_REPLICAS + 1) * MID_SYNC_STRIDE * sizeof(unsigned); }

int dihip_decode_mid_supported(int wbits, int hidden, int k_attn, int inter, int group_size) {
  alignas(16) static float dummy[4] = {0};
  GemvArgs a, b;
  int bo, bg, gpt;
  size_t lo, lg;
  if (!gemv_mid_plan(wbits, dummy, dummy, dummy, dummy, dummy, hidden, k_attn, dummy, 1e-6f, dummy, dummy, dummy, dummy, dummy, inter,
                     group_size, &a, &bo, &lo, &b, &bg, &lg, &gpt))
    return 0;
  int ncu = cached_num_cus();
  if (ncu <= 0) ncu = 256;
  // every workgroup of the launch must be resident at once (two per CU), and two of the larger kind must fit a CU's LDS
  return bo <= ncu && bg <= ncu && 2 * std::max(lo, lg) <= 128 * 1024 ? 1 : 0;
}

int dihip_decode_mid(void* stream, int wbits, const void* attn, const void* wo_packed, const void* wo_sz, const float* h_res,
                     float* h_out, const void* gamma, float eps, const void* wg_packed, const void* wg_sz, const void* wu_packed,
                     const void* wu_sz, void* act, int hidden, int k_attn, int inter, int group_size, void* sync, size_t sync_bytes,
                     int dtype) {
  DIHIP_REQUIRE(attn && wo_packed && wo_sz && h_out && gamma && wg_packed && wg_sz && wu_packed && wu_sz && act && sync, DIHIP_PARAM_ERROR,
                "decode_mid: null pointer");
  DIHIP_REQUIRE(dtype == DIHIP_BF16, DIHIP_PARAM_ERROR, "decode_mid: bf16 activations only");
  DIHIP_REQUIRE(sync_bytes >= dihip_decode_mid_sync_bytes(), DIHIP_MEMORY_ERROR, "decode_mid: sync buffer too small");
  DIHIP_REQUIRE(dihip_decode_mid_supported(wbits, hidden, k_attn, inter, group_size), DIHIP_PARAM_ERROR,
                "decode_mid: configuration not covered (one row, decode-GEMV shapes of at most one workgroup per CU each); see _supported");
  GemvArgs go, gg;
  int bo, bg, gpt;
  size_t lo, lg;
  DIHIP_REQUIRE(gemv_mid_plan(wbits, attn, wo_packed, wo_sz, h_res, h_out, hidden, k_attn, gamma, eps, wg_packed, wg_sz, wu_packed, wu_sz,
                              act, inter, group_size, &go, &bo, &lo, &gg, &bg, &lg, &gpt),
                DIHIP_PARAM_ERROR, "decode_mid: the GEMVs are not served by the decode GEMV (alignment)");
  unsigned* words = reinterpret_cast<unsigned*>(sync);
  static int presleep = -1;  // DIHIP_MID_PRESLEEP: s_sleep(127) repetitions (~3.9 us each) before a consumer's first poll
  if (presleep < 0) {
    const char* e = getenv("DIHIP_MID_PRESLEEP");
    presleep = e ? std::max(0, atoi(e)) : 0;
  }
  static int nowait = -1;  // DIHIP_MID_DEBUG=nowait: consumers do not wait (WRONG results: prices the hand-over itself)
  if (nowait < 0) {
    const char* e = getenv("DIHIP_MID_DEBUG");
    nowait = (e && std::string(e) == "nowait") ? 1 : 0;
  }
  for (GemvArgs* g : {&go, &gg}) {
    g->chain_counter = words;
    g->chain_done = words + (size_t)MID_REPLICAS * MID_SYNC_STRIDE;
    g->chain_replicas = MID_REPLICAS;
    g->chain_target = nowait ? 0 : bo;
    g->chain_consumers = bg;
    g->chain_presleep = presleep;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = std::max(lo, lg);
  hipError_t e = hipErrorInvalidValue;
  if (wbits == 4 && gpt) e = launch_mid<4, 1>(go, gg, bo, bg, lds, s);
  else if (wbits == 4) e = launch_mid<4, 0>(go, gg, bo, bg, lds, s);
  else if (wbits == 8 && gpt) e = launch_mid<8, 1>(go, gg, bo, bg, lds, s);
  else if (wbits == 8) e = launch_mid<8, 0>(go, gg, bo, bg, lds, s);
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "decode_mid: launch failed: %s", hipGetErrorString(e));
  return DIHIP_SUCCESS;
}

}  // extern "C"
