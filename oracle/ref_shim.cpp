// oracle/ref_shim.cpp -- C-ABI shim around the REFERENCE's own host reference loops.
//
// TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This file contains no reference
// code.  oracle/Makefile slices the plain-C++ host loops out of the reference's test
// sources *where they lie under /root/reference* into build intermediates under
// oracle/_ref/ (git-ignored, deleted after the compile) and this shim #includes them:
//
//   _ref/gemm_lowp_ref_slice.inc   <- tests/cpp/operator/cuda/operator_gemm_lowp_test.cpp
//        PackU8ToU4x2, ComputeQuantParam, CPU_Quant_Weight_PerC/SubC,
//        CPU_SubC_Ref, CPU_PerC_Ref, CPU_FP16W4_PerC_Ref            (:17-219)
//   _ref/prefill_ref_slice.inc     <- tests/cpp/kernel/cuda/kernel_mhaprefill_test.cpp
//        pefill_check_with_reference                                  (:117-323)
//
// plus the reference's host bfloat16 type (csrc/common/hie_bfloat16.hpp, included by -I).
// The result, oracle/_ref/libdashinfer_ref.so, is what tests use to pin the numpy
// restatement (oracle/gemm_ref.py, oracle/attention.py) and what bench.py may time as
// cpu_baseline.kind == "reference".
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "hie_bfloat16.hpp"

using bf16_t = hie::bfloat16;

namespace AS_UTEST {
#include "_ref/gemm_lowp_ref_slice.inc"
}  // namespace AS_UTEST

namespace allspark {
namespace cuda {
struct trivial_t {
  int batch, nhead, phead, seqlen;
};
}  // namespace cuda
}  // namespace allspark

namespace AS_UTEST_PREFILL {
#include "_ref/prefill_ref_slice.inc"
}  // namespace AS_UTEST_PREFILL

namespace {
template <typename T>
std::vector<T> from_f32(const float* p, size_t n) {
  std::vector<T> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = T(p[i]);
  return v;
}
template <typename T>
void to_f32(const std::vector<T>& v, float* p) {
  for (size_t i = 0; i < v.size(); ++i) p[i] = float(v[i]);
}
}  // namespace

extern "C" {

// C = FT(alpha * sum_k A[m,k] * (float(B[k,n]) - float(Z[k/G,n])) * float(S[k/G,n]))
// ft: 0 = float, 1 = bfloat16 (inputs are given as float and converted to FT first, as the
// reference test holds FT vectors).  group <= 0 -> per-channel (CPU_PerC_Ref).
int ref_gemm_a16w8(const float* A, const int8_t* B, const float* S, const float* Z, float* C,
                   uint32_t M, uint32_t N, uint32_t K, int group, float alpha, int ft) {
  const uint32_t G = group > 0 ? (K + group - 1) / group : 1;
  std::vector<int8_t> b(B, B + size_t(K) * N);
  if (ft == 0) {
    auto a = from_f32<float>(A, size_t(M) * K);
    auto s = from_f32<float>(S, size_t(G) * N);
    auto z = from_f32<float>(Z, size_t(G) * N);
    std::vector<float> c(size_t(M) * N);
    if (group > 0) AS_UTEST::CPU_SubC_Ref<float, int8_t>(a, b, s, z, c, M, N, K, group, alpha);
    else AS_UTEST::CPU_PerC_Ref<float, int8_t>(a, b, s, z, c, M, N, K, alpha);
    to_f32(c, C);
  } else {
    auto a = from_f32<bf16_t>(A, size_t(M) * K);
    auto s = from_f32<bf16_t>(S, size_t(G) * N);
    auto z = from_f32<bf16_t>(Z, size_t(G) * N);
    std::vector<bf16_t> c(size_t(M) * N);
    if (group > 0) AS_UTEST::CPU_SubC_Ref<bf16_t, int8_t>(a, b, s, z, c, M, N, K, group, alpha);
    else AS_UTEST::CPU_PerC_Ref<bf16_t, int8_t>(a, b, s, z, c, M, N, K, alpha);
    to_f32(c, C);
  }
  return 0;
}

// The reference's A16W4 test path: the u4 values are held unpacked as uint8 [K,N] and fed to
// CPU_SubC_Ref / CPU_PerC_Ref<FT, uint8_t> (operator_gemm_lowp_test.cpp TestGemmA16W4*).
int ref_gemm_a16w4_unpacked(const float* A, const uint8_t* B, const float* S, const float* Z,
                            float* C, uint32_t M, uint32_t N, uint32_t K, int group, float alpha,
                            int ft) {
  const uint32_t G = group > 0 ? (K + group - 1) / group : 1;
  std::vector<uint8_t> b(B, B + size_t(K) * N);
  if (ft == 0) {
    auto a = from_f32<float>(A, size_t(M) * K);
    auto s = from_f32<float>(S, size_t(G) * N);
    auto z = from_f32<float>(Z, size_t(G) * N);
    std::vector<float> c(size_t(M) * N);
    if (group > 0) AS_UTEST::CPU_SubC_Ref<float, uint8_t>(a, b, s, z, c, M, N, K, group, alpha);
    else AS_UTEST::CPU_PerC_Ref<float, uint8_t>(a, b, s, z, c, M, N, K, alpha);
    to_f32(c, C);
  } else {
    auto a = from_f32<bf16_t>(A, size_t(M) * K);
    auto s = from_f32<bf16_t>(S, size_t(G) * N);
    auto z = from_f32<bf16_t>(Z, size_t(G) * N);
    std::vector<bf16_t> c(size_t(M) * N);
    if (group > 0) AS_UTEST::CPU_SubC_Ref<bf16_t, uint8_t>(a, b, s, z, c, M, N, K, group, alpha);
    else AS_UTEST::CPU_PerC_Ref<bf16_t, uint8_t>(a, b, s, z, c, M, N, K, alpha);
    to_f32(c, C);
  }
  return 0;
}

// Packed per-channel u4 reference (CPU_FP16W4_PerC_Ref), float FT.
int ref_gemm_a16w4_perc_packed(const float* A, const uint8_t* Bpack, const float* S,
                               const float* Z, float* C, uint32_t M, uint32_t N, uint32_t K) {
  const uint32_t NP = (N + 1) / 2;
  auto a = from_f32<float>(A, size_t(M) * K);
  std::vector<uint8_t> b(Bpack, Bpack + size_t(K) * NP);
  auto s = from_f32<float>(S, N + 1);
  auto z = from_f32<float>(Z, N + 1);
  std::vector<float> c(size_t(M) * N + 1);
  AS_UTEST::CPU_FP16W4_PerC_Ref<float>(a, b, s, z, c, M, N, K, NP);
  std::memcpy(C, c.data(), sizeof(float) * size_t(M) * N);
  return 0;
}

void ref_pack_u8_to_u4x2(const uint8_t* data, uint8_t* pack, int N, int NPack, int K) {
  std::vector<uint8_t> d(data, data + size_t(K) * N), p(size_t(K) * NPack);
  AS_UTEST::PackU8ToU4x2(d, p, N, NPack, K);
  std::memcpy(pack, p.data(), p.size());
}

// Test-side quantiser (zero clamped), float FT. qbits 8 -> int8 range, 4 -> [0,15].
int ref_test_quant_weight(const float* W, float* Q, float* S, float* Z, uint32_t N, uint32_t K,
                          int group, int qbits) {
  const float qmax = qbits == 8 ? 127.f : 15.f, qmin = qbits == 8 ? -128.f : 0.f;
  const int G = group > 0 ? (K + group - 1) / group : 1;
  auto w = from_f32<float>(W, size_t(K) * N);
  std::vector<float> q(size_t(K) * N), s(size_t(G) * N), z(size_t(G) * N);
  if (group > 0) AS_UTEST::CPU_Quant_Weight_SubC<float, float>(w, q, s, z, N, K, group, G, qmax, qmin);
  else AS_UTEST::CPU_Quant_Weight_PerC<float, float>(w, q, s, z, N, K, qmax, qmin);
  to_f32(q, Q);
  to_f32(s, S);
  to_f32(z, Z);
  return 0;
}

// The reference's prefill checker: concat [batch, seqlen, 3, nhead, phead] f32,
// output [batch, seqlen, nhead, phead] f32.  Returns 1 when output passes.
int ref_prefill_check(const float* concat, const float* output, int batch, int seqlen, int nhead,
                      int phead, float alpha, int causal, float feps) {
  allspark::cuda::trivial_t p{batch, nhead, phead, seqlen};
  return AS_UTEST_PREFILL::pefill_check_with_reference<float>(p, concat, output, alpha,
                                                              causal != 0, feps)
             ? 1
             : 0;
}

}  // extern "C"
