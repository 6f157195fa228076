// mall_bench.cpp -- does a weight stream that is already resident in the 256 MiB Infinity Cache (MALL) arrive faster
// than one that comes from HBM?  Decides whether a background weight prefetch (next kernels' weights pulled on-die
// while the latency-bound kernels of the decode chain run) can pay.
//   (1) bandwidth vs working set: every CU re-reads a private slice of an N-MiB buffer R times inside one launch
//       (register ring, 8 x 1 KiB per wave in flight, default-policy and nt loads);
//   (2) single cold pass vs single pass after a touch-prefetch (one dword per 128-byte line) of the same buffer,
//       rotating over enough buffers to defeat the cache.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mall_bench tools/mall_bench.cpp && ./tools/mall_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ u32x4 ld(const u32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// every wave owns `chunks` consecutive 1 KiB chunks and reads them `passes` times
template <bool NT>
__global__ __launch_bounds__(512) void stream_kernel(u32x4* __restrict__ out, const u32x4* __restrict__ w, long chunks, int passes) {
  constexpr int D = 8;
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const u32x4* p = w + gw * chunks * 64 + lane;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int ps = 0; ps < passes; ++ps) {
    u32x4 ring[D];
#pragma unroll
    for (int j = 0; j < D; ++j) ring[j] = ld<NT>(p + (long)(j < chunks ? j : chunks - 1) * 64);
    for (long c = 0; c < chunks; c += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        acc ^= ring[j];
        const long nxt = c + j + D < chunks ? c + j + D : chunks - 1;
        __builtin_amdgcn_sched_barrier(0);
        ring[j] = ld<NT>(p + nxt * 64);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) acc ^= ring[j];
  }
  if (acc[0] == 0x12345678u) out[gw * 64 + lane] = acc;
}

// touch one dword per 128-byte line: pulls the lines through L2 into the memory-side cache
__global__ __launch_bounds__(256) void touch_kernel(unsigned* __restrict__ out, const unsigned* __restrict__ w, long lines) {
  const long stride = (long)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < lines; i += stride) acc ^= w[i * 32];
  if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t big = 2048ull << 20;
  unsigned char* buf;
  CK(hipMalloc(&buf, big));
  CK(hipMemset(buf, 1, big));
  u32x4* out;
  CK(hipMalloc(&out, 64 << 20));
  const int blocks = 256, waves = 8;
  printf("# (1) re-read bandwidth vs working set (256 WGs x 8 waves, ring 8 KiB per wave)\n");
  for (int nt = 0; nt < 2; ++nt) {
    for (int mb : {8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024}) {
      const long chunks = ((long)mb << 20) / 1024 / (blocks * waves);
      const int passes = std::max(2, (int)(4096 / mb));
      auto go = [&]() {
        if (nt) hipLaunchKernelGGL(stream_kernel<true>, dim3(blocks), dim3(waves * 64), 0, st, out, (const u32x4*)buf, chunks, passes);
        else hipLaunchKernelGGL(stream_kernel<false>, dim3(blocks), dim3(waves * 64), 0, st, out, (const u32x4*)buf, chunks, passes);
      };
      go();
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      go();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = (double)chunks * 1024 * blocks * waves * passes;
      printf("  %s  %5d MiB x %4d passes: %8.1f us  %7.0f GB/s\n", nt ? "nt     " : "default", mb, passes, ms * 1e3, bytes / ms / 1e6);
    }
  }
  printf("# (2) one pass over a 72 MiB buffer: cold (rotating over 24 buffers) vs after a touch-prefetch of the same buffer\n");
  const size_t one = 72ull << 20;
  const int nb = (int)(big / one);
  const long chunks = (long)(one / 1024 / (blocks * waves));
  for (int nt = 0; nt < 2; ++nt) {
    auto pass = [&](int b) {
      if (nt) hipLaunchKernelGGL(stream_kernel<true>, dim3(blocks), dim3(waves * 64), 0, st, out, (const u32x4*)(buf + b * one), chunks, 1);
      else hipLaunchKernelGGL(stream_kernel<false>, dim3(blocks), dim3(waves * 64), 0, st, out, (const u32x4*)(buf + b * one), chunks, 1);
    };
    auto touch = [&](int b, int tb) {
      hipLaunchKernelGGL(touch_kernel, dim3(tb), dim3(256), 0, st, (unsigned*)out, (const unsigned*)(buf + b * one), (long)(one / 128));
    };
    for (int r = 0; r < 2; ++r)
      for (int b = 0; b < nb; ++b) pass(b);
    CK(hipStreamSynchronize(st));
    float ms_cold, ms_touch, ms_both;
    CK(hipEventRecord(e0, st));
    for (int b = 0; b < nb; ++b) pass(b);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_cold, e0, e1));
    for (int tb : {256, 1024, 2048}) {
      CK(hipEventRecord(e0, st));
      for (int b = 0; b < nb; ++b) touch(b, tb);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_touch, e0, e1));
      CK(hipEventRecord(e0, st));
      for (int b = 0; b < nb; ++b) {
        touch(b, tb);
        pass(b);
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_both, e0, e1));
      printf("  %s  cold pass %6.2f us (%5.0f GB/s) | touch (%4d WGs) %6.2f us (%5.0f GB/s) | pass after touch %6.2f us (%5.0f GB/s)\n",
             nt ? "nt     " : "default", ms_cold * 1e3 / nb, one / (ms_cold / nb) / 1e6, tb, ms_touch * 1e3 / nb, one / (ms_touch / nb) / 1e6,
             (ms_both - ms_touch) * 1e3 / nb, one / ((ms_both - ms_touch) / nb) / 1e6);
    }
    // the same with the pass lagging the touch by one / two buffers (touch b+lag, pass b): is it still resident?
    for (int lag : {1, 2}) {
      CK(hipEventRecord(e0, st));
      for (int b = 0; b < nb; ++b) {
        touch((b + lag) % nb, 1024);
        pass(b);
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms_both, e0, e1));
      printf("  %s  touch(b+%d) ; pass(b): %6.2f us per pair (touch alone %6.2f, cold pass %6.2f)\n", nt ? "nt     " : "default", lag,
             ms_both * 1e3 / nb, ms_touch * 1e3 / nb, ms_cold * 1e3 / nb);
    }
  }
  // (3) two streams: touch kernels on a side stream run concurrently with the passes on the main stream
  printf("# (3) side-stream prefetch: stream B touches buffer b+1 while stream A passes over buffer b (events between them)\n");
  hipStream_t sb;
  CK(hipStreamCreate(&sb));
  std::vector<hipEvent_t> ev(nb + 2);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st));
    for (int b = 0; b < nb; ++b) {
      // B: touch b+1 (no dependency on A); A: wait for the touch of b (recorded last iteration), then pass b
      hipLaunchKernelGGL(touch_kernel, dim3(64), dim3(256), 0, sb, (unsigned*)out, (const unsigned*)(buf + ((b + 1) % nb) * one), (long)(one / 128));
      CK(hipEventRecord(ev[b + 1], sb));
      if (b > 0) CK(hipStreamWaitEvent(st, ev[b], 0));
      hipLaunchKernelGGL(stream_kernel<true>, dim3(blocks), dim3(waves * 64), 0, st, out, (const u32x4*)(buf + b * one), chunks, 1);
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  eager two-stream: %6.2f us per buffer (%5.0f GB/s)\n", ms * 1e3 / nb, one / (ms / nb) / 1e6);
  }
  return 0;
}
