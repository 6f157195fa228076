// oracle/ref_attn_shim.cpp -- C-ABI shim around the REFERENCE's own x86 decoder attention (pin the decode
// attention oracle).  TEST INFRASTRUCTURE ONLY; this file contains no reference code.  oracle/Makefile slices, where they lie under
// /root/reference, into build intermediates under oracle/_ref/ (git-ignored, deleted after the compile):
//   _ref/batch_mqa_dec.inc    <- csrc/core/operator/generate_opt/batch_mqa/batch_mqa_op.cpp   cpu_dec_single_mqa          (:140-179)
//   _ref/mha_x86_softmax.inc  <- csrc/core/kernel/cpu/mha.cpp   the AVX2 vSoftmax / vLogSoftmax / vSoftmaxMask             (:28-373)
//   _ref/mha_decode.inc       <- csrc/core/kernel/cpu/mha.cpp   UpdateKVLauncher, GetBatchArrayLauncher,
//                                MultiQueryGetBatchArrayLauncher, BatchGemmWraper<float>, BatchSoftmax<float>              (:709-806)
// and this shim supplies what they lean on and cannot be built here: the declarations of cpu_kernel.h for those templates, a
// serial parallel_for (cpu_common.h:127-146 is OpenMP / TBB), DispatchCPU for FLOAT32, and cblas_sgemm -- MKL is an LFS stub in
// the reference tree, so the matrix products run through a plain row-major triple loop with float accumulation in k
// order: the reference's score / softmax / P.V STRUCTURE, pointer arithmetic, GQA head mapping, cache update and its own AVX2
// softmax (polynomial exp) are the reference's; the k-summation order of the two products stays unpinned (stated in oracle/attention.py).
#include <immintrin.h>

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

enum CBLAS_LAYOUT { CblasRowMajor = 101, CblasColMajor = 102 };
enum CBLAS_TRANSPOSE { CblasNoTrans = 111, CblasTrans = 112 };
static void cblas_sgemm(CBLAS_LAYOUT layout, CBLAS_TRANSPOSE ta, CBLAS_TRANSPOSE tb, int m, int n, int k, float alpha, const float* A, int lda,
                        const float* B, int ldb, float beta, float* C, int ldc) {
  assert(layout == CblasRowMajor);
  (void)layout;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      float acc = 0.f;
      for (int p = 0; p < k; ++p) {
        const float a = ta == CblasNoTrans ? A[(size_t)i * lda + p] : A[(size_t)p * lda + i];
        const float b = tb == CblasNoTrans ? B[(size_t)p * ldb + j] : B[(size_t)j * ldb + p];
        acc += a * b;
      }
      C[(size_t)i * ldc + j] = alpha * acc + (beta == 0.f ? 0.f : beta * C[(size_t)i * ldc + j]);
    }
}

namespace allspark {
enum DataType { DATATYPE_UNDEFINED = 0, FLOAT32 = 1 };
class DeviceContext {};
template <typename F>
void DispatchCPU(DataType dtype, F&& f) {
  assert(dtype == FLOAT32);
  (void)dtype;
  f.template operator()<float>();
}
namespace cpu {
template <typename T0, typename F>
void parallel_for(const T0& D0, const F& func) {
  for (T0 d0 = 0; d0 < D0; ++d0) func(d0);
}
// cpu_kernel.h: the primary templates the sliced specialisations belong to
template <typename T>
void LogSoftmaxKernel(const T* input, T* output, int outer_dim, int inner_dim);
template <typename T>
void SoftmaxKernel(T* input, int* len_arr, int outer_dim, int inner_dim, float temperature = 1.0);
template <typename T>
void UpdateKVLauncher(T* k, T* v, const T* step_k, const T* step_v, int batch_size, int step, int max_length, int hidden_size, int seq_len,
                      int stride);
template <typename T>
void GetBatchArrayLauncher(T* q, T* k, T* v, T* score, T* out, T** q_array, T** k_array, T** v_array, T** score_array, T** out_array,
                           int batch_size, int beam_size, int num_heads, int size_per_head, int step, int q_stride, int kv_stride,
                           int score_stride, int out_stride);
template <typename T>
void MultiQueryGetBatchArrayLauncher(T* q, T* k, T* v, T* score, T* out, T** q_array, T** k_array, T** v_array, T** score_array, T** out_array,
                                     int batch_size, int beam_size, int num_heads, int size_per_head, int group_num, int step, int q_stride,
                                     int kv_stride, int score_stride, int out_stride);
template <typename T>
void BatchGemmWraper(void** matrix_C, void** matrix_A, void** matrix_B, int m, int n, int k, bool transA, bool transB, float alpha, float beta,
                     int lda, int ldb, int ldc, int batch);
template <typename T>
void BatchSoftmax(T* score, const float* mask, int batch_size, int beam_size, int num_heads, int seq_len, int step);
template <typename T>
void SimpleAdd(T* out, const T* in1, const T* in2, int count) {
  for (int i = 0; i < count; ++i) out[i] = in1[i] + in2[i];
}
#include "_ref/mha_x86_softmax.inc"
#include "_ref/mha_decode.inc"
}  // namespace cpu
#include "_ref/batch_mqa_dec.inc"
}  // namespace allspark

extern "C" {

// One decoder step of BatchMQAOp's x86 path for `batch` requests that all sit at the same step (the op's own restriction):
// qkv [batch, (n + 2 g) H] f32 rows (q | k | v of this step, already rotated), k_cache / v_cache [batch, cache_max_len, g H] f32
// holding step - 1 earlier tokens; the step's k / v are appended at position step - 1 (UpdateKVLauncher), attention runs over
// `step` tokens.  out [batch, n H].
int ref_dec_single_mqa(float* out, const float* qkv, float* k_cache, float* v_cache, int batch, int step, int cache_max_len, int n, int H,
                       int g, float alpha) {
  const int hidden = n * H, kv_stride = g * H;
  const int gemm_batch = batch * n;
  std::vector<float> score((size_t)batch * n * step);
  std::vector<void*> qa(gemm_batch), ka(gemm_batch), va(gemm_batch), sa(gemm_batch), oa(gemm_batch);
  const float* key = qkv + hidden;
  const float* value = key + kv_stride;
  allspark::DeviceContext ctx;
  allspark::cpu_dec_single_mqa(allspark::FLOAT32, out, score.data(), qkv, key, value, nullptr, nullptr, k_cache, v_cache, qa.data(), ka.data(),
                               va.data(), sa.data(), oa.data(), batch, 1, 1, step, cache_max_len, hidden, n, H, g, gemm_batch, alpha, nullptr,
                               0, &ctx);
  return 0;
}

// the reference's AVX2 softmax on one row (in place)
void ref_vsoftmax(float* row, int n, float temperature) { allspark::cpu::vSoftmax(n, row, temperature); }

}  // extern "C"
