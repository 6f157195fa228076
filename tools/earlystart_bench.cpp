// earlystart_bench.cpp -- experiment for DESIGN.md section 8, item 1.  Run at the end of round 3 (one MI355X, 56 dependent stages):
//     MB per stage   plain boundaries   early start (two streams)   pure stream at 6.7 TB/s
//          8              4.17 us              21.0 us                     1.25 us
//         16              5.32                 36.1                        2.50
//         36              8.27                 39.4                        5.63
//         72             13.75                 46.7                       11.27
// -> the two-stream early start is a dead end (each hand-over between the streams costs 17-30 us); the "boundaries" column is the
// floor of a chain of bare streaming kernels: 2.5-2.9 us per stage on top of the bytes.  The product's four GEMVs per layer (8.8, 6.4,
// 72 and 36 MB: 5.44 + 4.31 + 15.05 + 9.29 = 34.1 us) sit within 13 % of that chain (4.2 + 4.0 + 13.75 + 8.27 = 30.2 us).
//
// Question: a decode layer is a chain of dependent weight-streaming kernels, and each launch spends 3-5 us not
// streaming (ramp, first-byte latency, tail) -- `tools/ldsdma_bench` shows the transport itself reaches 6.4-6.9 TB/s.
// Can the NEXT kernel start before its input exists?  Its weight stream does not depend on the previous kernel's
// output, only its activations do.  Here a "stage" stands for one GEMV: it requests its weight ring at once, THEN waits
// for the previous stage's completion counter (the stand-in for "my activations are ready"), streams its bytes (XOR
// consumer) and bumps its own counter.
//   mode 0: one stream, plain kernel boundaries (what the hipGraph replay does today);
//   mode 1: two streams, stage i on stream i % 2, dependency through the counters: stage i + 1 is resident, its ring in
//           flight and its workgroups parked on the counter while stage i streams.
// Both stages fit a CU together (8 waves each, no LDS, few registers), so every workgroup of both is resident and the
// spin cannot starve the producer; the spin is bounded anyway (give-up flag) so that a mistake cannot hang the box.
//   hipcc --offload-arch=gfx950 -O3 -o tools/earlystart_bench tools/earlystart_bench.cpp && timeout 60 ./tools/earlystart_bench
#include <hip/hip_runtime.h>

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
constexpr int D = 8;          // ring depth: chunks of 1 KiB per wave in flight
constexpr int WAVES = 8;      // per workgroup

__global__ __launch_bounds__(WAVES * 64) void stage(u32x4* __restrict__ out, const u32x4* __restrict__ w, long chunks_per_wave,
                                                     const unsigned* flag_in, unsigned expected, unsigned* flag_out,
                                                     unsigned* gave_up) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * WAVES + (threadIdx.x >> 6);
  const u32x4* p = w + gw * chunks_per_wave * 64 + lane;
  // 1. the weight ring leaves immediately: it needs nothing from the previous stage
  u32x4 ring[D];
#pragma unroll
  for (int j = 0; j < D; ++j) ring[j] = __builtin_nontemporal_load(p + (long)(j < chunks_per_wave ? j : chunks_per_wave - 1) * 64);
  // 2. "activations ready": one lane per workgroup polls the producer's counter (device scope), bounded
  if (flag_in) {
    if (threadIdx.x == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(flag_in, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expected) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) {  // ~ seconds: give up loudly instead of hanging
          atomicAdd(gave_up, 1u);
          break;
        }
      }
    }
    __syncthreads();
  }
  // 3. stream
  u32x4 acc = {0u, 0u, 0u, 0u};
  long c = 0;
  for (; c + D <= chunks_per_wave; c += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      acc ^= ring[j];
      const long nxt = c + j + D < chunks_per_wave ? c + j + D : chunks_per_wave - 1;
      __builtin_amdgcn_sched_barrier(0);
      ring[j] = __builtin_nontemporal_load(p + nxt * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  out[gw * 64 + lane] = acc;
  // 4. done: one release-increment per workgroup
  if (flag_out) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag_out, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int nstages = 56;                                        // two "layers" of 28
  const size_t mb[] = {8, 16, 36, 72};                           // stage sizes of the Qwen2-7B int4 layer (qkv / o .. gate+up)
  hipStream_t s[2];
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  unsigned *flags, *gave_up;
  CK(hipMalloc(&flags, (nstages + 1) * sizeof(unsigned)));
  CK(hipMalloc(&gave_up, sizeof(unsigned)));
  u32x4* out;
  CK(hipMalloc(&out, (size_t)ncu * WAVES * 64 * 16));
  printf("%d CUs; chain of %d dependent streaming stages; us per stage\n", ncu, nstages);
  for (size_t m : mb) {
    const size_t bytes = m << 20;
    const long cpw = (long)(bytes / 1024) / ((long)ncu * WAVES);
    // distinct buffers per stage (3.5 GB at 72 MB would be the real layer set; 16 buffers keep the Infinity Cache cold enough)
    const int nbuf = 16;
    std::vector<u32x4*> w(nbuf);
    for (auto& p : w) {
      CK(hipMalloc(&p, bytes));
      CK(hipMemset(p, 0x5a, bytes));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms[2] = {0.f, 0.f};
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(flags, 0, (nstages + 1) * sizeof(unsigned), s[0]));
        CK(hipMemsetAsync(gave_up, 0, sizeof(unsigned), s[0]));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s[0]));
        if (mode == 1) CK(hipStreamWaitEvent(s[1], e0, 0));
        for (int i = 0; i < nstages; ++i) {
          hipStream_t st = mode == 0 ? s[0] : s[i & 1];
          const unsigned* fin = (mode == 1 && i > 0) ? flags + (i - 1) : nullptr;
          unsigned* fout = mode == 1 ? flags + i : nullptr;
          hipLaunchKernelGGL(stage, dim3(ncu), dim3(WAVES * 64), 0, st, out, w[i % nbuf], cpw, fin, (unsigned)ncu, fout, gave_up);
        }
        if (mode == 1) {  // join: the timed region ends when both streams are drained
          hipEvent_t j;
          CK(hipEventCreate(&j));
          CK(hipEventRecord(j, s[1]));
          CK(hipStreamWaitEvent(s[0], j, 0));
        }
        CK(hipEventRecord(e1, s[0]));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[mode], e0, e1));
      }
    }
    unsigned hg = 0;
    CK(hipMemcpy(&hg, gave_up, sizeof(unsigned), hipMemcpyDeviceToHost));
    const double ideal = (double)bytes / 6.7e12 * 1e6;
    printf("  %3zu MB/stage: boundaries %7.2f us   early start %7.2f us   (pure stream at 6.7 TB/s: %5.2f us)%s\n", m,
           ms[0] * 1e3 / nstages, ms[1] * 1e3 / nstages, ideal, hg ? "   [SPIN GAVE UP: early-start number invalid]" : "");
    for (auto& p : w) CK(hipFree(p));
  }
  return 0;
}
