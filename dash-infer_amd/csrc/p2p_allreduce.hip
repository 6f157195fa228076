// p2p_allreduce.hip -- one-shot peer-to-peer sum all-reduce for the tensor-parallel decode step
// (include/dashinfer_hip.h section 6b).
//
// The decode step all-reduces one hidden row per request (Qwen2-7B: 7 KB) twice per layer
// (AllReduceOp after o_proj and down_proj, csrc/core/operator/nccl/allreduce/allreduce_op.cpp:84-92).  At that size a
// ring collective is pure latency: 2 (n - 1) dependent xGMI hops.  xGMI is point-to-point -- every GPU has a direct
// link to every other GPU of the node -- so here every rank WRITES its row straight into a slot of every peer's receive
// buffer (one hop, all links in parallel), raises a flag next to it, waits for the n - 1 flags addressed to itself and
// sums the n rows locally, in rank order, in f32: one launch, one hop, and bit-identical results on every rank (which
// the replicated greedy sampling of the decoder relies on; RCCL's ring gives the same guarantee).
//
// Protocol (per call = "epoch" e, parity p = e & 1):
//   push   workgroup w of rank r copies piece w of its input into slot[p][r] of EVERY rank's buffer (its own included)
//          with system-scope stores, fences (system), then stores e into flag[p][r][w] of every rank's buffer;
//   wait   workgroup w of rank q polls flag[p][r][w] of ITS OWN buffer for all r until they read >= e;
//   sum    it adds piece w of slot[p][0..n-1] of its own buffer (system-scope loads: the lines were written by peers
//          over xGMI, and may sit stale in this GPU's caches from epoch e - 2) and writes the result.
// Two parities suffice: a peer can start epoch e + 2 (which overwrites slot[p]) only after it has finished epoch e + 1,
// which needs this rank's epoch-(e + 1) push, which this rank issues after its epoch-e kernel has completed.
// The epoch lives in device memory and is advanced by the last workgroup of each launch, so a captured launch replays
// correctly (nothing in the kernel arguments changes from call to call).
//
// The receive buffer (slots + flags) is FINE-GRAINED / uncached device memory (hipExtMallocWithFlags, as RCCL allocates its
// own flag and LL buffers): peers write it over xGMI while the owner's kernel polls and reads it in the same launch, and
// coarse-grained memory (plain hipMalloc) is only coherent at kernel boundaries -- the owner's L2 may keep a line of
// epoch e - 2 whatever scope the load names.  An allocation that cannot be made that way is an error, never a
// silent coarse-grained substitute.
//
// Bounded: messages up to P2P_MAX_BYTES; a poll that sees no progress for tens of seconds traps (a lost peer must not hang the
// GPU silently).  Developed on a one-GPU box: ranks as host threads / processes on one GPU exercise the protocol
// (tests/test_gpu_tp_loopback.py, test_gpu_p2p_processes.py); it has NOT run across xGMI yet -- bench.py therefore verifies
// every all-reduce of its warm-up steps against RCCL before it times this path (decoder.P2PComm.verify).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "device_utils.h"

using namespace dihip;

namespace {

constexpr int P2P_MAX_RANKS = 8;
constexpr int P2P_MAX_WGS = 32;
constexpr size_t P2P_MAX_BYTES = 256 * 1024;  // per rank and call (batch 32 x hidden 3584 x bf16 = 229 KB)
constexpr int P2P_THREADS = 256;
constexpr size_t P2P_FLAG_STRIDE = 64;  // one flag per 64-byte line
constexpr size_t P2P_DATA_BYTES = 2 * P2P_MAX_RANKS * P2P_MAX_BYTES;
constexpr size_t P2P_FLAGS_BYTES = 2 * P2P_MAX_RANKS * P2P_MAX_WGS * P2P_FLAG_STRIDE;
constexpr size_t P2P_BUFFER_BYTES = P2P_DATA_BYTES + P2P_FLAGS_BYTES;

struct P2PState {  // device-resident, private to the rank
  unsigned epoch;  // completed calls
  unsigned done;   // workgroups of the running call that have finished
  unsigned error;  // a wait gave up (probe mode: dihip_p2p_ar_set_timeout(.., trap = 0)); the communicator is unusable afterwards
};

struct P2PComm {
  int rank, nranks;
  unsigned char* bufs[P2P_MAX_RANKS];
  P2PState* state;
  unsigned long long max_spins = 1ull << 24;  // tens of seconds
  int trap = 1;
};

struct P2PArgs {
  unsigned char* bufs[P2P_MAX_RANKS];
  P2PState* state;
  unsigned long long max_spins;
  int trap;
  const void* in;
  void* out;
  unsigned count;  // elements
  int rank, nranks;
};

__device__ __forceinline__ unsigned char* slot_ptr(unsigned char* buf, unsigned parity, int src) {
  return buf + ((size_t)parity * P2P_MAX_RANKS + src) * P2P_MAX_BYTES;
}
__device__ __forceinline__ unsigned* flag_ptr(unsigned char* buf, unsigned parity, int src, int wg) {
  return reinterpret_cast<unsigned*>(buf + P2P_DATA_BYTES + (((size_t)parity * P2P_MAX_RANKS + src) * P2P_MAX_WGS + wg) * P2P_FLAG_STRIDE);
}

template <int FT>
__global__ __launch_bounds__(P2P_THREADS) void p2p_allreduce_kernel(const P2PArgs a) {
  constexpr int ES = FT == DIHIP_F32 ? 4 : 2;
  constexpr int EPW = 8 / ES;  // elements per 8-byte word
  __shared__ unsigned s_epoch;
  const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  if (tid == 0) s_epoch = __hip_atomic_load(&a.state->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  __syncthreads();
  const unsigned epoch = s_epoch, parity = epoch & 1u;
  // this workgroup's piece, in 8-byte words (the message is padded to whole words by the host contract: count % EPW == 0)
  const unsigned words = a.count / EPW;
  const unsigned per = (words + nwg - 1) / nwg;
  const unsigned w0 = min(words, wg * per), w1 = min(words, w0 + per);
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.in);
  // ---- push: my piece into my slot of every rank's buffer (peers first, myself last)
  for (unsigned i = w0 + tid; i < w1; i += P2P_THREADS) {
    const unsigned long long v = src[i];
    for (int d = 1; d <= a.nranks; ++d) {
      const int peer = (a.rank + d) % a.nranks;
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(slot_ptr(a.bufs[peer], parity, a.rank));
      __hip_atomic_store(dst + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the stores above are visible before the flags below
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < a.nranks) {
    const int peer = (a.rank + 1 + tid) % a.nranks;
    __hip_atomic_store(flag_ptr(a.bufs[peer], parity, a.rank, wg), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- wait: the flags addressed to me for this piece
  if (tid < a.nranks) {
    const unsigned* f = flag_ptr(a.bufs[a.rank], parity, tid, wg);
    unsigned long long spins = 0;
    // signed distance: flags only move forward, a flag "ahead" of this epoch cannot occur for this parity
    while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > a.max_spins) {  // without the peer's row for that long: fail loudly, never hang the box
        if (a.trap) __builtin_trap();
        __hip_atomic_store(&a.state->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // probe mode: report and leave
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // ---- sum in rank order (f32), write the result
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out);
  for (unsigned i = w0 + tid; i < w1; i += P2P_THREADS) {
    float acc[EPW];
#pragma unroll
    for (int e = 0; e < EPW; ++e) acc[e] = 0.f;
    for (int r = 0; r < a.nranks; ++r) {
      const unsigned long long* sl = reinterpret_cast<const unsigned long long*>(slot_ptr(a.bufs[a.rank], parity, r));
      const unsigned long long v = __hip_atomic_load(sl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if constexpr (FT == DIHIP_F32) {
        acc[0] += __uint_as_float((unsigned)v);
        acc[1] += __uint_as_float((unsigned)(v >> 32));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += ft_bits_to_f32<FT>((unsigned)(v >> (16 * e)) & 0xFFFFu);
      }
    }
    unsigned long long o = 0;
    if constexpr (FT == DIHIP_F32) {
      o = (unsigned long long)__float_as_uint(acc[0]) | ((unsigned long long)__float_as_uint(acc[1]) << 32);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o |= (unsigned long long)f32_to_ft_bits<FT>(acc[e]) << (16 * e);
    }
    dst[i] = o;
  }
  // ---- the last workgroup of the launch advances the epoch (every workgroup read it before arriving here)
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(&a.state->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)nwg - 1u) {
      __hip_atomic_store(&a.state->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.state->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

extern "C" {

size_t dihip_p2p_ar_buffer_bytes(void) { return P2P_BUFFER_BYTES; }
size_t dihip_p2p_ar_max_bytes(void) { return P2P_MAX_BYTES; }

int dihip_p2p_ar_alloc(void** buf) {
  DIHIP_REQUIRE(buf, DIHIP_PARAM_ERROR, "p2p_ar_alloc: null pointer");
  void* p = nullptr;
  // its own allocation (an IPC handle names a whole allocation), uncached: never held in this GPU's L2, so a peer's write
  // is what the next load returns.  Fine-grained is the fallback spelling of the same property on runtimes without the
  // uncached flag.  DIHIP_P2P_COARSE=1 (diagnostics on a one-GPU box only) takes plain hipMalloc.
  hipError_t e = hipErrorUnknown;
  const char* coarse = getenv("DIHIP_P2P_COARSE");
  if (coarse && coarse[0] == '1') {
    e = hipMalloc(&p, P2P_BUFFER_BYTES);
  } else {
    e = hipExtMallocWithFlags(&p, P2P_BUFFER_BYTES, hipDeviceMallocUncached);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipExtMallocWithFlags(&p, P2P_BUFFER_BYTES, hipDeviceMallocFinegrained);
    }
  }
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_MEMORY_ERROR, "p2p_ar_alloc: uncached / fine-grained device allocation failed: %s", hipGetErrorString(e));
  e = hipMemset(p, 0, P2P_BUFFER_BYTES);
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "p2p_ar_alloc: hipMemset: %s", hipGetErrorString(e));
  *buf = p;
  return DIHIP_SUCCESS;
}

int dihip_p2p_ar_free(void* buf) {
  if (buf) (void)hipFree(buf);
  return DIHIP_SUCCESS;
}

int dihip_ipc_get_handle(void* dev_ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "dihip_ipc_*: the handle travels as 64 bytes");
  DIHIP_REQUIRE(dev_ptr && handle64, DIHIP_PARAM_ERROR, "ipc_get_handle: null pointer");
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, dev_ptr);
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set)",
                hipGetErrorString(e));
  std::memcpy(handle64, &h, 64);
  return DIHIP_SUCCESS;
}

int dihip_ipc_open_handle(const void* handle64, void** dev_ptr) {
  DIHIP_REQUIRE(dev_ptr && handle64, DIHIP_PARAM_ERROR, "ipc_open_handle: null pointer");
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, 64);
  void* p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
  *dev_ptr = p;
  return DIHIP_SUCCESS;
}

int dihip_ipc_close_handle(void* dev_ptr) {
  if (dev_ptr) (void)hipIpcCloseMemHandle(dev_ptr);
  return DIHIP_SUCCESS;
}

int dihip_p2p_ar_create(void** comm, int rank, int nranks, void* const* bufs) {
  DIHIP_REQUIRE(comm && bufs && nranks >= 1 && nranks <= P2P_MAX_RANKS && rank >= 0 && rank < nranks, DIHIP_PARAM_ERROR,
                "p2p_ar_create: bad argument (1..%d ranks)", P2P_MAX_RANKS);
  P2PComm* c = new P2PComm{};
  c->rank = rank;
  c->nranks = nranks;
  for (int r = 0; r < nranks; ++r) {
    if (!bufs[r]) {
      delete c;
      set_last_error("p2p_ar_create: buffer of rank %d is null", r);
      return DIHIP_PARAM_ERROR;
    }
    c->bufs[r] = reinterpret_cast<unsigned char*>(bufs[r]);
  }
  hipError_t e = hipMalloc(&c->state, sizeof(P2PState));
  if (e == hipSuccess) e = hipMemset(c->state, 0, sizeof(P2PState));
  if (e != hipSuccess) {
    delete c;
    set_last_error("p2p_ar_create: state allocation: %s", hipGetErrorString(e));
    return DIHIP_MEMORY_ERROR;
  }
  *comm = c;
  return DIHIP_SUCCESS;
}

int dihip_p2p_ar_destroy(void* comm) {
  if (!comm) return DIHIP_SUCCESS;
  P2PComm* c = reinterpret_cast<P2PComm*>(comm);
  (void)hipFree(c->state);
  delete c;
  return DIHIP_SUCCESS;
}

int dihip_p2p_allreduce_sum(void* comm, void* stream, const void* in, void* out, size_t count, int dtype) {
  DIHIP_REQUIRE(comm && in && out, DIHIP_PARAM_ERROR, "p2p_allreduce: null pointer");
  DIHIP_REQUIRE(dtype == DIHIP_F32 || dtype == DIHIP_F16 || dtype == DIHIP_BF16, DIHIP_PARAM_ERROR, "p2p_allreduce: dtype %d", dtype);
  if (count == 0) return DIHIP_SUCCESS;
  const size_t es = dtype == DIHIP_F32 ? 4 : 2;
  DIHIP_REQUIRE(count * es <= P2P_MAX_BYTES, DIHIP_EXCEED_LIMIT_ERROR, "p2p_allreduce: %zu bytes exceed the %zu-byte slot (use RCCL)",
                count * es, P2P_MAX_BYTES);
  DIHIP_REQUIRE((count * es) % 8 == 0 && (reinterpret_cast<uintptr_t>(in) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0,
                DIHIP_PARAM_ERROR, "p2p_allreduce: message must be whole 8-byte words, 8-byte aligned");
  P2PComm* c = reinterpret_cast<P2PComm*>(comm);
  P2PArgs a{};
  for (int r = 0; r < c->nranks; ++r) a.bufs[r] = c->bufs[r];
  a.state = c->state;
  a.max_spins = c->max_spins;
  a.trap = c->trap;
  a.in = in;
  a.out = out;
  a.count = (unsigned)count;
  a.rank = c->rank;
  a.nranks = c->nranks;
  // one workgroup per 4 KB of message: a decode row (7 KB) takes two; every rank derives the same grid from the count
  const unsigned words = (unsigned)(count * es / 8);
  const int wgs = (int)std::min<unsigned>(P2P_MAX_WGS, std::max(1u, (words + 2 * P2P_THREADS - 1) / (2 * P2P_THREADS)));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DIHIP_BF16) hipLaunchKernelGGL(p2p_allreduce_kernel<DIHIP_BF16>, dim3(wgs), dim3(P2P_THREADS), 0, s, a);
  else if (dtype == DIHIP_F16) hipLaunchKernelGGL(p2p_allreduce_kernel<DIHIP_F16>, dim3(wgs), dim3(P2P_THREADS), 0, s, a);
  else hipLaunchKernelGGL(p2p_allreduce_kernel<DIHIP_F32>, dim3(wgs), dim3(P2P_THREADS), 0, s, a);
  return launch_status();
}

int dihip_p2p_ar_set_timeout(void* comm, unsigned long long max_spins, int trap) {
  DIHIP_REQUIRE(comm && max_spins > 0, DIHIP_PARAM_ERROR, "p2p_ar_set_timeout: bad argument");
  P2PComm* c = reinterpret_cast<P2PComm*>(comm);
  c->max_spins = max_spins;
  c->trap = trap ? 1 : 0;
  return DIHIP_SUCCESS;
}

int dihip_p2p_ar_error(void* comm, int* error) {
  DIHIP_REQUIRE(comm && error, DIHIP_PARAM_ERROR, "p2p_ar_error: null pointer");
  P2PComm* c = reinterpret_cast<P2PComm*>(comm);
  unsigned e = 0;
  const hipError_t rc = hipMemcpy(&e, &c->state->error, sizeof(e), hipMemcpyDeviceToHost);  // synchronises with the device
  DIHIP_REQUIRE(rc == hipSuccess, DIHIP_RUNTIME_ERROR, "p2p_ar_error: hipMemcpy: %s", hipGetErrorString(rc));
  *error = (int)e;
  return DIHIP_SUCCESS;
}

}  // extern "C"
