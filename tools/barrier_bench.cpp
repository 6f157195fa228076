// barrier_bench.cpp -- cost of an in-kernel grid barrier (256 WGs x 512 threads, 1 per CU) on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* err) {
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0; ok = 1;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { ok = 0; atomicExch(err, 1u); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok != 0;
}

// XCD-hierarchical: per-XCD counters (blocks b%8), leaders combine on a top counter
__device__ __forceinline__ bool grid_barrier_xcd(unsigned* ctr, unsigned gen, unsigned nblocks, unsigned* err) {
  // ctr[0..7*16]: per-xcd arrive counters (64B apart), ctr[128]: top counter, ctr[144 + x*16]: per-xcd generation
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = 1;
    const unsigned x = blockIdx.x & 7, per = (nblocks + 7 - x) / 8;  // blocks with b%8==x
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t = __hip_atomic_fetch_add(ctr + x * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    if (t == gen * per + per - 1) {  // last of this XCD
      __hip_atomic_fetch_add(ctr + 128, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (gen + 1) * 8) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { ok = 0; atomicExch(err, 1u); break; }
      }
      __hip_atomic_store(ctr + 144 + x * 16, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(ctr + 144 + x * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen + 1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { ok = 0; atomicExch(err, 1u); break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok != 0;
}

__global__ __launch_bounds__(512) void k_barriers(unsigned* ctr, unsigned* err, float* data, int nbar, int mode, int work) {
  float acc = 0.f;
  for (int i = 0; i < nbar; ++i) {
    if (work) {  // publish a little data each phase so the release has something to write back
      data[(size_t)blockIdx.x * 512 + threadIdx.x] = acc + i;
    }
    bool ok = mode == 0 ? grid_barrier(ctr, (unsigned)(i + 1) * gridDim.x, err) : grid_barrier_xcd(ctr, (unsigned)i, gridDim.x, err);
    if (!ok) return;
    if (work) acc += data[(size_t)((blockIdx.x + 1) % gridDim.x) * 512 + threadIdx.x];
  }
  if (acc == 12345.f) data[0] = acc;
}

int main() {
  unsigned *ctr, *err; float* data;
  CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&data, 256 * 512 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int work = 0; work < 2; ++work)
      for (int nbar : {1, 101}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipMemsetAsync(ctr, 0, 4096, st)); CK(hipMemsetAsync(err, 0, 4, st));
          CK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(k_barriers, dim3(256), dim3(512), 0, st, ctr, err, data, nbar, mode, work);
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        static float base[2][2];
        if (nbar == 1) base[mode][work] = best;
        else printf("mode %s work %d: %.2f us per barrier (kernel with 1 barrier %.1f us)%s\n", mode ? "xcd-hier" : "flat-counter", work,
                    (best - base[mode][work]) * 1e3 / 100, base[mode][work] * 1e3, herr ? "  TIMEOUT" : "");
      }
  return 0;
}
