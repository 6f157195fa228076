// gemm_skeleton_bench.cpp -- what matrix-pipe utilisation can a GEMM SKELETON reach on gfx950 before any arithmetic detail?
//
// Round 3 left the context-phase GEMM (gemm_prefill_kernel.hpp) at ~40 % of the matrix pipe and showed with a timing-only
// build that the integer expansion / fix-up is not what the rest goes to: the skeleton is (two lock-stepped waves per SIMD,
// one LDS fragment read per two MFMAs, one barrier per k-tile).  This tool times candidate inner loops with the operands
// already in LDS / registers (no global traffic, no epilogue), so that the next kernel is chosen by measurement:
//
//   A  "2w"   the round-3 skeleton: 8 waves (2 per SIMD), wave tile 128 x 32 (8 x 2 accumulators), per k-step 8 A fragments
//             from LDS, B fragments in registers, one barrier per k-tile of 4 k-steps
//   B  "1w"   one wave per SIMD (4 waves, up to 512 registers): wave tile 128 x 128 (8 x 8 accumulators), per k-step 8 A + 8 B
//             fragments from LDS (16 reads per 64 MFMAs), one barrier per k-tile
//   C  "1w-p" B with the fragment reads of k-step s + 1 issued before the MFMAs of k-step s (software pipelined)
//   D  "2w-p" A with the same pipelining
//
// Output: TFLOP/s and the fraction of 2.5 PFLOP/s per variant; `clock` from wall time vs the kernel's own cycle counter.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gemm_skeleton_bench tools/gemm_skeleton_bench.cpp && ./tools/gemm_skeleton_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);     \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

__device__ __forceinline__ f32x4_t mfma(const u32x4_t& a, const u32x4_t& b, const f32x4_t& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

constexpr int KSTEPS = 4;  // k-steps (32 k) per k-tile

// ---- A / D: 8 waves, wave tile 8 row tiles x 2 column tiles, A from LDS (fragment-major), B in registers ----------------
template <bool PIPE>
__global__ __launch_bounds__(512) void skel_2w(float* out, int ktiles, unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x [8 row tiles][KSTEPS][64 lanes] x 16 B
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4_t* lds = reinterpret_cast<u32x4_t*>(smem);
  for (int i = threadIdx.x; i < 2 * 8 * KSTEPS * 64; i += 512) lds[i] = u32x4_t{0x3F803F80u + i, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  __syncthreads();
  f32x4_t acc[8][2];
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 2; ++c) acc[r][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t b[2][KSTEPS];
  for (int c = 0; c < 2; ++c)
    for (int k = 0; k < KSTEPS; ++k) b[c][k] = u32x4_t{0x3F803F80u, 0x3F803F80u + (unsigned)(wave + c + k), 0x3F803F80u, 0x3F803F80u};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int kt = 0; kt < ktiles; ++kt) {
    const u32x4_t* a = lds + (kt & 1) * 8 * KSTEPS * 64 + lane;
    if constexpr (!PIPE) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const u32x4_t af = a[(r * KSTEPS + ks) * 64];
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[r][c] = mfma(af, b[c][ks], acc[r][c]);
        }
      }
    } else {
      u32x4_t cur[8], nxt[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) cur[r] = a[(r * KSTEPS + 0) * 64];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        if (ks + 1 < KSTEPS) {
#pragma unroll
          for (int r = 0; r < 8; ++r) nxt[r] = a[(r * KSTEPS + ks + 1) * 64];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[r][c] = mfma(cur[r], b[c][ks], acc[r][c]);
#pragma unroll
        for (int r = 0; r < 8; ++r) cur[r] = nxt[r];
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 2; ++c) s += acc[r][c][0] + acc[r][c][1] + acc[r][c][2] + acc[r][c][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// ---- B / C: 4 waves (one per SIMD), wave tile 8 x 8 accumulators, A and B fragments from LDS --------------------------------
template <bool PIPE, int KS>
__global__ __launch_bounds__(256) void skel_1w(float* out, int ktiles, unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 stages x ([16 row tiles] + [16 column tiles]) x [KS][64 lanes] x 16 B
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves over a 256 x 256 workgroup tile
  u32x4_t* lds = reinterpret_cast<u32x4_t*>(smem);
  constexpr int STAGE = 32 * KS * 64;  // u32x4 per stage: 16 A row tiles + 16 B column tiles
  for (int i = threadIdx.x; i < 2 * STAGE; i += 256) lds[i] = u32x4_t{0x3F803F80u + i, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  __syncthreads();
  f32x4_t acc[8][8];
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 8; ++c) acc[r][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int kt = 0; kt < ktiles; ++kt) {
    const u32x4_t* a = lds + (kt & 1) * STAGE + (wm * 8) * KS * 64 + lane;
    const u32x4_t* bb = lds + (kt & 1) * STAGE + (16 + wn * 8) * KS * 64 + lane;
    if constexpr (!PIPE) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4_t bf[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) bf[c] = bb[(c * KS + ks) * 64];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const u32x4_t af = a[(r * KS + ks) * 64];
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[r][c] = mfma(af, bf[c], acc[r][c]);
        }
      }
    } else {
      u32x4_t ac[8], bc[8], an[8], bn[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ac[i] = a[(i * KS + 0) * 64];
        bc[i] = bb[(i * KS + 0) * 64];
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            an[i] = a[(i * KS + ks + 1) * 64];
            bn[i] = bb[(i * KS + ks + 1) * 64];
          }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[r][c] = mfma(ac[r], bc[c], acc[r][c]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ac[i] = an[i];
          bc[i] = bn[i];
        }
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 8; ++c) s += acc[r][c][0] + acc[r][c][1] + acc[r][c][2] + acc[r][c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, int threads, size_t lds, double flops_per_wg_ktile, int wgs, int ktiles) {
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)wgs * threads * 4));
  CK(hipMalloc(&cyc, 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, 0, out, ktiles, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), lds, 0, out, ktiles, cyc);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c = 0;
  CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double sec = ms * 1e-3 / reps;
  const double tf = flops_per_wg_ktile * wgs * ktiles / sec / 1e12;
  printf("%-6s %4d workgroups x %3d threads, %6d k-tiles: %8.1f us  %7.1f TFLOP/s = %4.1f %% of 2.5 PF;  %.0f cycles per k-tile in workgroup 0 (%.2f GHz)\n",
         name, wgs, threads, ktiles, sec * 1e6, tf, tf / 25.0, (double)c / ktiles, (double)c / sec / 1e9);
  CK(hipFree(out));
  CK(hipFree(cyc));
}

int main(int argc, char** argv) {
  const int ktiles = argc > 1 ? atoi(argv[1]) : 2000;
  int dev = 0, cus = 256;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const size_t lds2w = (size_t)2 * 8 * KSTEPS * 64 * 16;  // 64 KB: two A stages of 128 rows x 128 k
  // one workgroup per CU everywhere (the 2w kernel's 244 registers and the 1w kernel's tile allow no more)
  run("2w", skel_2w<false>, 512, lds2w, 2.0 * 128 * 256 * 32 * KSTEPS, cus, ktiles);
  run("2w-p", skel_2w<true>, 512, lds2w, 2.0 * 128 * 256 * 32 * KSTEPS, cus, ktiles);
  // 1w: (16 + 16) tiles x 1 KB per k-step; k-tiles of 2 k-steps (64 k) keep two stages at 128 KB of the 160 KB LDS
  constexpr int KS1 = 2;
  const size_t lds1w = (size_t)2 * 32 * KS1 * 64 * 16;
  run("1w", skel_1w<false, KS1>, 256, lds1w, 2.0 * 256 * 256 * 32 * KS1, cus, 2 * ktiles);
  run("1w-p", skel_1w<true, KS1>, 256, lds1w, 2.0 * 256 * 256 * 32 * KS1, cus, 2 * ktiles);
  return 0;
}
