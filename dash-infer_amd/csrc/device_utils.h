// device_utils.h -- shared helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dashinfer_hip.h"

namespace dihip {

// ---------------------------------------------------------------- host side ----------------
void set_last_error(const char* fmt, ...);

#define DIHIP_CHECK_HIP(expr, code)                                                   \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      dihip::set_last_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,              \
                            hipGetErrorString(_e));                                   \
      return (code);                                                                  \
    }                                                                                 \
  } while (0)

#define DIHIP_REQUIRE(cond, code, ...)                                                \
  do {                                                                                \
    if (!(cond)) {                                                                    \
      dihip::set_last_error(__VA_ARGS__);                                             \
      return (code);                                                                  \
    }                                                                                 \
  } while (0)

inline int launch_status(int code = DIHIP_RUNTIME_ERROR) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error("kernel launch failed: %s", hipGetErrorString(e));
    return code;
  }
  return 0;
}

int cached_num_cus();
// diagnostics: the buffer set by dihip_debug_set_trace if it holds at least `bytes`, else null
unsigned long long* debug_trace_buffer(size_t bytes);
// environment switches: read once into a function-local `static const` (C++11 static initialisation is thread-safe; the rank
// threads of a loop-back TP group used to race on hand-rolled `static int x = -1` caches)
int env_int(const char* name, int dflt);          // atoi of the variable, or dflt when unset
bool env_off(const char* name);                   // set and starting with '0'
void debug_set_trace(void* buf, size_t bytes);

// ---------------------------------------------------------------- device side --------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
// round-to-nearest-even, NaN quieted (same as torch / the reference's host bf16)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
  _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)b);
  return (float)h;
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;  // v_cvt_f16_f32: RNE
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}

// FT codes: DIHIP_F32=0, DIHIP_F16=1, DIHIP_BF16=2
template <int FT>
__device__ __forceinline__ float ft_bits_to_f32(uint32_t b) {
  if constexpr (FT == DIHIP_BF16) return bf16_bits_to_f32(b);
  else return f16_bits_to_f32(b);
}
template <int FT>
__device__ __forceinline__ uint32_t f32_to_ft_bits(float f) {
  if constexpr (FT == DIHIP_BF16) return f32_to_bf16_bits(f);
  else return f32_to_f16_bits(f);
}
// two f32 -> two packed FT (round to nearest even): ONE v_cvt_pk_bf16_f32 on gfx950 for bf16
template <int FT>
__device__ __forceinline__ uint32_t pack_ft2(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ v = {a, b};
  if constexpr (FT == DIHIP_BF16) {
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));
  } else {
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_));
  }
}
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) { return pack_ft2<DIHIP_BF16>(a, b); }
template <int FT>
__device__ __forceinline__ float ft_round(float f) {
  return ft_bits_to_f32<FT>(f32_to_ft_bits<FT>(f));
}
template <int FT>
__device__ __forceinline__ float load_ft(const void* p, size_t i) {
  if constexpr (FT == DIHIP_F32) return ((const float*)p)[i];
  else return ft_bits_to_f32<FT>(((const uint16_t*)p)[i]);
}
template <int FT>
__device__ __forceinline__ void store_ft(void* p, size_t i, float v) {
  if constexpr (FT == DIHIP_F32) ((float*)p)[i] = v;
  else ((uint16_t*)p)[i] = (uint16_t)f32_to_ft_bits<FT>(v);
}

// ds_read_b64_tr_b16: a 16-lane group reads a 4 x 16 block of 16-bit elements TRANSPOSED -- lane p supplies the address
// of row p/4, columns (p%4)*4 .. +4, and receives column p of the block (4 rows).  Pinned by tools/trread_test.cpp.
__device__ __forceinline__ u32x2_t lds_read_tr16(const unsigned char* p) {
  typedef short v4i16_ __attribute__((ext_vector_type(4)));
  const v4i16_ r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16_*)(p));
  return __builtin_bit_cast(u32x2_t, r);
}

// FRAG32 activation layout (DIHIP_ACT_FRAG32): the 16-bit matrix x[M, K] stored as the MFMA A fragments the
// small-batch kernel consumes -- [K/32 k-steps][MT 16-row tiles][lane = kb*16 + row][8 elements], so that one
// fragment is ONE contiguous 1 KiB wave-load (row-major x makes it 16 pieces of 64 B from 16 rows: half-used
// cache lines and 16 tag look-ups per load; measured 1.2-1.7x slower kernels).  MT = 1 for M <= 16, else 2.
__host__ __device__ inline size_t act_frag_index(int m, int k, int mt) {
  return ((((size_t)(k >> 5) * mt + (m >> 4)) * 64 + ((k & 31) >> 3) * 16 + (m & 15)) * 8 + (k & 7));
}

// Load through a pointer that was itself read from memory (span tables): hipcc cannot tell its address space
// and emits FLAT loads, which count on vmcnt AND lgkmcnt and may return out of order -- every wait on one
// becomes s_waitcnt vmcnt(0) lgkmcnt(0) and drains all prefetches.  The explicit global address space turns
// them into global_load with exact in-order vmcnt accounting.
template <typename T>
__device__ __forceinline__ T gload(const void* p) {
  return *(const __attribute__((address_space(1))) T*)(p);
}
template <typename T>
__device__ __forceinline__ void gstore(void* p, const T& v) {
  *(__attribute__((address_space(1))) T*)(p) = v;
}

// Reductions over the 4 rows of 16 lanes (lane bits 4 and 5), result in every lane: v_permlane32_swap / v_permlane16_swap
// (gfx950) exchange half-waves / odd-even rows between two registers -- VALU only, no LDS round trip as ds_bpermute.
__device__ __forceinline__ float rows_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const unsigned um = __float_as_uint(m);
  const auto r2 = __builtin_amdgcn_permlane16_swap(um, um, false, false);
  return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}
__device__ __forceinline__ float rows_sum(float v) {  // (r0 + r1) + (r2 + r3): the order of two xor-16 / xor-32 shuffles
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned um = __float_as_uint(m);
  const auto r2 = __builtin_amdgcn_permlane32_swap(um, um, false, false);
  return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}
// Sum over the 64 lanes, result in every lane.  DPP moves inside the 16-lane rows (pairs, quads, 8s, 16s: after each level
// every lane of a group holds the group's sum, so which lane of the partner group is read does not matter), then the two
// cross-row exchanges of rows_sum: VALU only.  (__shfl_xor is six dependent ds_bpermute round trips through the LDS pipe:
// ~0.25 us of the RMSNorm prologue every decode GEMV runs.)  One fixed tree: every norm path of the library sums through here.
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float v) {  // DPP move, all rows / banks enabled
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov_f32<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f32<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f32<0x141>(v);  // row_half_mirror: the other quad of the 8
  v += dpp_mov_f32<0x140>(v);  // row_mirror: the other half of the 16
  return rows_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {  // same moves as wave_sum (max is order-independent: same bits as any tree)
  v = fmaxf(v, dpp_mov_f32<0xB1>(v));
  v = fmaxf(v, dpp_mov_f32<0x4E>(v));
  v = fmaxf(v, dpp_mov_f32<0x141>(v));
  v = fmaxf(v, dpp_mov_f32<0x140>(v));
  return rows_max(v);
}
__device__ __forceinline__ float wave_min(float v) { return -wave_max(-v); }  // min(a, b) = -max(-a, -b) bit for bit (no NaNs here)

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {  // allspark UnaryType
    case DIHIP_ACT_TANH: return tanhf(v);
    case DIHIP_ACT_GELU_ERF: return 0.5f * v * (1.f + erff(v * 0.70710678f));
    case DIHIP_ACT_GELU_TANH: return 0.5f * v * (1.f + tanhf(0.7978845608f * (v + 0.044715f * v * v * v)));
    case DIHIP_ACT_RELU: return fmaxf(v, 0.f);
    case DIHIP_ACT_SILU: return v / (1.f + expf(-v));
    case DIHIP_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// sin/cos of pos * inv_freq: the f32 product (as the reference computes it, rotary.cpp:22-106) is
// range-reduced in f64 so that sincosf stays on its fast small-argument path (the large-argument
// Payne-Hanek path of the device libm spills to scratch).
__device__ __forceinline__ void rope_sincos(uint32_t pos, float inv_freq, float* sn, float* cs) {
  const float ang = (float)pos * inv_freq;
  const double two_pi = 6.283185307179586476925286766559;
  double r = (double)ang;
  r -= two_pi * floor(r / two_pi);
  if (r > 3.14159265358979323846) r -= two_pi;
  sincosf((float)r, sn, cs);
}

// In-launch split hand-off (cdna_hip_programming.md G16, counter form): every wave of the
// block has finished its plain slab stores; returns true in ALL threads of exactly one block
// per counter -- the last to arrive -- after an agent-scope acquire, so that it may read the
// other blocks' slabs with plain loads.  `flag_lds` is one word of the block's single LDS array.
__device__ __forceinline__ bool arrive_and_check_last(unsigned* counter, unsigned expected,
                                                      unsigned* flag_lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = (t == expected - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      // self-reset: the counter is zero again for the next launch (graph replay safe)
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *flag_lds = last ? 1u : 0u;
  }
  __syncthreads();
  return *flag_lds != 0u;
}

}  // namespace dihip
