// ldsdma_bench.cpp -- weight-stream transport experiment for the decode GEMV:
// how fast can every CU pull a private, contiguous byte stream out of HBM when the chunks land
//   (a) in a REGISTER ring  (global_load_dwordx4, D x 16 B per lane in flight: what gemv_stream_kernel does), or
//   (b) in an LDS ring via LDS-DMA (global_load_lds_dwordx4: no VGPRs held by bytes in flight, ring depth bounded by
//       the 160 KB of LDS instead of the register file), consumed with ds_read_b128?
// Each wave owns a contiguous region and walks it in 1 KiB chunks (64 lanes x 16 B, the packed-weight chunk of the
// GEMV kernels); the "consumer" is an XOR over the chunk so that nothing is optimised away.  Waits are counted
// (s_waitcnt vmcnt(D - 1)): the oldest chunk is consumed while D - 1 newer ones stay in flight.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ldsdma_bench tools/ldsdma_bench.cpp && ./tools/ldsdma_bench
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

// (a) register ring: D chunks of 16 B per lane in flight, consume-then-refill, fully unrolled ring
template <int D>
__global__ __launch_bounds__(1024) void stream_regs(u32x4* __restrict__ out, const u32x4* __restrict__ w, long chunks_per_wave) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const u32x4* p = w + gw * chunks_per_wave * 64 + lane;
  u32x4 ring[D];
#pragma unroll
  for (int j = 0; j < D; ++j) ring[j] = __builtin_nontemporal_load(p + (long)j * 64);
  u32x4 acc = {0u, 0u, 0u, 0u};
  long c = 0;
  for (; c + D <= chunks_per_wave; c += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      acc ^= ring[j];
      const long nxt = c + j + D < chunks_per_wave ? c + j + D : chunks_per_wave - 1;  // clamped: unconditional loads
      __builtin_amdgcn_sched_barrier(0);
      ring[j] = __builtin_nontemporal_load(p + nxt * 64);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  out[gw * 64 + lane] = acc;
}

// (b) LDS ring through LDS-DMA.  The builtin issues global_load_lds_dwordx4 (M0 = LDS base of the slot, lane-linear
// 1 KiB image); the reads are inline asm so that hipcc does not put its conservative vmcnt(0) in front of them.
template <int D>
__global__ __launch_bounds__(1024) void stream_ldsdma(u32x4* __restrict__ out, const u32x4* __restrict__ w, long chunks_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long gw = (long)blockIdx.x * (blockDim.x >> 6) + wave;
  const u32x4* p = w + gw * chunks_per_wave * 64 + lane;
  unsigned char* ring = smem + (size_t)wave * D * 1024;
  auto issue = [&](int slot, long chunk) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + chunk * 64),
                                     (__attribute__((address_space(3))) void*)(ring + slot * 1024), 16, 0, 2 /* nt */);
  };
#pragma unroll
  for (int j = 0; j < D; ++j) issue(j, j);
  u32x4 acc = {0u, 0u, 0u, 0u};
  const unsigned lds_lane = (unsigned)(size_t)(ring) + lane * 16;  // LDS byte address of this lane's 16 B in slot 0
  long c = 0;
  for (; c + D <= chunks_per_wave; c += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      u32x4 v;
      asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                   : "=v"(v)
                   : "v"(lds_lane + j * 1024), "n"(D - 1)
                   : "memory");
      acc ^= v;
      const long nxt = c + j + D < chunks_per_wave ? c + j + D : chunks_per_wave - 1;
      issue(j, nxt);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[gw * 64 + lane] = acc;
}

template <typename K>
static double run(K kernel, int blocks, int waves, size_t lds, u32x4* out, const u32x4* w, long cpw, size_t total_bytes, int reps) {
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(waves * 64), lds, 0, out, w, cpw);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(waves * 64), lds, 0, out, w, cpw);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)total_bytes * reps / (ms * 1e-3) / 1e9;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const size_t total = (size_t)1 << 30;  // 1 GiB streamed per launch: far beyond L2 + Infinity Cache reuse between launches? (256 MB MALL: 3/4 misses)
  u32x4 *w, *out;
  CK(hipMalloc(&w, total));
  CK(hipMemset(w, 0x5a, total));
  CK(hipMalloc(&out, (size_t)ncu * 16 * 64 * 16 * 2));
  printf("%d CUs; GB/s of a 1 GiB stream (one workgroup per CU), by waves per workgroup and KiB in flight per wave\n", ncu);
  for (int waves : {4, 8, 16}) {
    const long cpw = (long)(total / 1024) / ((long)ncu * waves);
    const size_t bytes = (size_t)cpw * 1024 * ncu * waves;
    printf("waves %2d | regs: D=4 %6.0f  D=8 %6.0f  D=12 %6.0f", waves, run(stream_regs<4>, ncu, waves, 0, out, w, cpw, bytes, 5),
           run(stream_regs<8>, ncu, waves, 0, out, w, cpw, bytes, 5), run(stream_regs<12>, ncu, waves, 0, out, w, cpw, bytes, 5));
    printf(" | lds-dma: D=4 %6.0f  D=8 %6.0f", run(stream_ldsdma<4>, ncu, waves, (size_t)waves * 4 * 1024, out, w, cpw, bytes, 5),
           run(stream_ldsdma<8>, ncu, waves, (size_t)waves * 8 * 1024, out, w, cpw, bytes, 5));
    if (waves * 16 <= 160) printf("  D=16 %6.0f", run(stream_ldsdma<16>, ncu, waves, (size_t)waves * 16 * 1024, out, w, cpw, bytes, 5));
    if (waves * 32 <= 160) printf("  D=32 %6.0f", run(stream_ldsdma<32>, ncu, waves, (size_t)waves * 32 * 1024, out, w, cpw, bytes, 5));
    printf("\n");
  }
  return 0;
}
