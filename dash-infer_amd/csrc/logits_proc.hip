// logits_proc.hip -- the two parts of GenerateOp around the sampling itself, over f32 logits:
//
//  1. the logits processors that run BEFORE sampling (generate_op.cpp:536-538 process_logits_launcher ->
//     generate_impl_gpu.hpp:94-111 gen_process_logits_gpu -> cuda::LogitsProcessor, kernel/cuda/beam_search.cu:456-539), in the
//     reference's order, per request (the BatchGencfg lists, generate_op.cpp:239-312):
//       repetition penalty    beam_search.cu:330-357  for every DISTINCT token t among ids[lo .. cur_len) (lo = input_len when
//                             suppress_repetition_in_generation, else 0): s = s < 0 ? s * p : s / p -- the reference reads a copy of
//                             the scores, so a token that occurs twice is penalised once
//       frequency / presence  beam_search.cu:359-392  count[t] over the GENERATED ids[input_len .. cur_len):
//                             s -= count * frequency + (count > 0 ? presence : 0)
//       no-repeat n-gram      beam_search.cu:394-420  a position whose n-1 tokens equal the last n-1 tokens bans the token that followed
//                             it (s = -1e9)
//       minimum length        beam_search.cu:423-433  cur_len < min_length: s[eos] = -1e9
//     The reference spends six launches, a copy of the [batch, vocab] scores and a memset of a [batch, vocab] count array per step.
//     Here: ONE launch, one workgroup per request, driven by the request's TOKENS -- it touches O(cur_len) logits, not O(vocab): the
//     count array is scratch whose touched entries the kernel zeroes itself (no memset, no state between calls), and "each distinct
//     token once" is decided by the atomic that counts it.  Results are bit-identical to LogitsProcessor<float> evaluated without
//     contraction (-ffp-contract=off: count * frequency and + presence are two roundings; oracle/logits_proc.py).
//
//  2. log-probabilities AFTER sampling (generate_op.cpp:600-606 logprobs_launcher -> generate_impl_gpu.hpp:33-80 logprobs_gpu:
//     log-softmax of the processed logits, the chosen token's value, the top `top_logprobs` <= 10 values and indices): one launch, one
//     workgroup per request; the [batch, vocab] log-probability tensor is never written -- the row's maximum and log-sum-exp, then
//     `top_n` selection rounds in (value descending, index ascending) order.
#include "device_utils.h"
#include "dashinfer_hip.h"

namespace dihip {

constexpr int LP_THREADS = 1024;
constexpr int LP_MAX_TOP = 32;

struct LogitsProcArgs {
  float* logits;        // [M, N], processed in place
  const int64_t* ids;   // [M, max_len]: prompt + generated tokens of every request (max_dec_ids, generate_op.cpp:521-534), or
  int64_t* const* rows; // [M] per-request device-resident histories of max_len ids (a null row: nothing to process) when ids == null
  const int64_t* append;  // rows form: the step's input id of every request, written to history[cur_len - 1] first (may be null)
  int max_len, N;
  const int* cur_len;   // [M] tokens in ids[m] (step + in_length_bias)
  const int* input_len; // [M] prompt length
  const float* repetition;  // [M] (1: off)
  const float* frequency;   // [M] (0: off)
  const float* presence;    // [M] (0: off)
  const int* ngram;     // [M] no_repeat_ngram_size (0: off)
  const int* min_length;  // [M]
  const int* eos;       // [M]
  const int* suppress;  // [M] suppress_repetition_in_generation
  int* count;           // [M, N] scratch
};

__global__ __launch_bounds__(LP_THREADS) void logits_processor_kernel(const LogitsProcArgs a) {
  const int row = blockIdx.x, tid = threadIdx.x;
  float* s = a.logits + (size_t)row * a.N;
  int* cnt = a.count + (size_t)row * a.N;
  const int cur_raw = a.cur_len[row];
  const int L = min(max(cur_raw, 0), a.max_len);
  const int64_t* ids;
  if (a.ids) {
    ids = a.ids + (size_t)row * a.max_len;
  } else {
    int64_t* h = a.rows[row];
    if (!h) return;  // (uniform over the workgroup)
    // the device-resident history grows by the step's input id before it is read: the decode step stays a pure function of device memory
    if (a.append && L >= 1 && L == cur_raw) {
      if (tid == 0) h[L - 1] = a.append[row];
      __threadfence();
      __syncthreads();
    }
    ids = h;
  }
  const int in_len = a.input_len[row];
  const int lo_rep = a.suppress[row] != 0 ? max(in_len, 0) : 0;
  const int lo_gen = max(in_len, 0);
  const float p = a.repetition[row], fq = a.frequency[row], pr = a.presence[row];
  auto valid = [&](int64_t t) { return t >= 0 && t < (int64_t)a.N; };

  for (int i = tid; i < L; i += LP_THREADS) {
    const int64_t t = ids[i];
    if (valid(t)) cnt[t] = 0;
  }
  __threadfence();
  __syncthreads();
  // repetition penalty: the lane that counts a token first applies it (the score it reads is the unprocessed one)
  for (int i = lo_rep + tid; i < L; i += LP_THREADS) {
    const int64_t t = ids[i];
    if (valid(t) && atomicAdd(&cnt[t], 1) == 0) {
      const float v = s[t];
      s[t] = v < 0.f ? v * p : v / p;
    }
  }
  __threadfence();
  __syncthreads();
  for (int i = tid; i < L; i += LP_THREADS) {
    const int64_t t = ids[i];
    if (valid(t)) cnt[t] = 0;
  }
  __threadfence();
  __syncthreads();
  // frequency / presence: count the generated tokens, then the lane that claims a token (bit 30) applies its count
  for (int i = lo_gen + tid; i < L; i += LP_THREADS) {
    const int64_t t = ids[i];
    if (valid(t)) atomicAdd(&cnt[t], 1);
  }
  __threadfence();
  __syncthreads();
  for (int i = lo_gen + tid; i < L; i += LP_THREADS) {
    const int64_t t = ids[i];
    if (!valid(t)) continue;
    const int old = atomicOr(&cnt[t], 1 << 30);
    if (old >> 30) continue;
    float total = (float)old * fq;   // token_count[tid] * frequency_penalty_list[batch]
    if (old > 0) total = total + pr;
    s[t] = s[t] - total;
  }
  __threadfence();
  __syncthreads();
  // no-repeat n-gram (cur_len as given: the comparison window ends at the last token)
  const int ng = a.ngram[row];
  if (ng > 0) {
    for (int i = tid; i < L; i += LP_THREADS) {
      if (i + ng - 2 < L - 1) {
        bool same = true;
        for (int j = 0; j < ng - 1 && same; ++j) same = ids[i + j] == ids[L - ng + j + 1];
        if (same) {
          const int64_t t = ids[i + ng - 1];
          if (valid(t)) s[t] = -1e9f;
        }
      }
    }
  }
  if (tid == 0 && cur_raw < a.min_length[row]) {
    const int e = a.eos[row];
    if (e >= 0 && e < a.N) s[e] = -1e9f;
  }
}

struct LogprobArgs {
  const float* logits;   // [M, N]
  const int64_t* chosen; // [M] sampled tokens (may be null)
  int N, top_n, out_stride;
  float* token_logprob;  // [M] (may be null)
  float* top_value;      // [M, out_stride]
  int* top_index;        // [M, out_stride]
  // records form (top_value == null): row m writes {token_logprob, top values [out_stride], top indices [out_stride]} = 1 + 2 * out_stride
  // words at records[m] + (position[m] + position_bias) * (1 + 2 * out_stride); a null records[m]: the request did not ask
  float* const* records;
  const uint32_t* position;
  int position_bias, max_records;
};

__device__ __forceinline__ uint32_t lp_order_key(float v) {  // larger float <=> larger key; NaN below everything (as csrc/sample.hip)
  uint32_t b = __float_as_uint(v);
  if ((b & 0x7FFFFFFFu) > 0x7F800000u) return 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// 64-bit sort key: (value key, ~index) -- the larger key is the better candidate (value descending, index ascending)
__device__ __forceinline__ uint64_t lp_pack(float v, int idx) { return ((uint64_t)lp_order_key(v) << 32) | (uint32_t)(~(uint32_t)idx); }

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(LP_THREADS) void logprobs_kernel(const LogprobArgs a) {
  __shared__ float red_f[LP_THREADS / 64];
  __shared__ uint64_t red_k[LP_THREADS / 64];
  __shared__ float bcast_f[2];
  __shared__ uint64_t bcast_k;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = a.logits + (size_t)row * a.N;
  float* tok_out = a.token_logprob ? a.token_logprob + row : nullptr;
  float* val_out = a.top_value ? a.top_value + (size_t)row * a.out_stride : nullptr;
  int* idx_out = a.top_index ? a.top_index + (size_t)row * a.out_stride : nullptr;
  if (a.records) {
    float* rec = a.records[row];
    const long long p = (long long)(a.position ? a.position[row] : 0u) + a.position_bias;
    if (!rec || p < 0 || p >= a.max_records) return;  // (uniform over the workgroup)
    rec += (size_t)p * (1 + 2 * a.out_stride);
    tok_out = rec;
    val_out = rec + 1;
    idx_out = reinterpret_cast<int*>(rec + 1 + a.out_stride);
  }
  // row maximum
  float mx = -INFINITY;
  for (int i = tid; i < a.N; i += LP_THREADS) mx = fmaxf(mx, x[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) red_f[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    float m = red_f[0];
    for (int w = 1; w < LP_THREADS / 64; ++w) m = fmaxf(m, red_f[w]);
    bcast_f[0] = m;
  }
  __syncthreads();
  mx = bcast_f[0];
  // sum of exp(x - max): per-thread partial sums in double (a fixed order: thread stride, wave butterfly, waves in order)
  double sum = 0.0;
  for (int i = tid; i < a.N; i += LP_THREADS) sum += (double)expf(x[i] - mx);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  __shared__ double red_d[LP_THREADS / 64];
  if (lane == 0) red_d[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < LP_THREADS / 64; ++w) t += red_d[w];
    bcast_f[1] = (float)log(t);
  }
  __syncthreads();
  const float lse = bcast_f[1];
  if (tid == 0 && tok_out && a.chosen) {
    const int64_t c = a.chosen[row];
    *tok_out = (c >= 0 && c < (int64_t)a.N) ? (x[c] - mx) - lse : -INFINITY;
  }
  // top_n: round r takes the best key strictly below the previous round's
  uint64_t below = ~0ull;
  for (int r = 0; r < a.top_n; ++r) {
    uint64_t best = 0ull;
    for (int i = tid; i < a.N; i += LP_THREADS) {
      const uint64_t k = lp_pack(x[i], i);
      if (k < below && k > best) best = k;
    }
    best = wave_max_u64(best);
    if (lane == 0) red_k[wave] = best;
    __syncthreads();
    if (tid == 0) {
      uint64_t b = red_k[0];
      for (int w = 1; w < LP_THREADS / 64; ++w) b = red_k[w] > b ? red_k[w] : b;
      bcast_k = b;
      const int idx = (int)(~(uint32_t)(b & 0xFFFFFFFFu));
      const bool any = b != 0ull && idx >= 0 && idx < a.N;
      idx_out[r] = any ? idx : -1;
      val_out[r] = any ? (x[idx] - mx) - lse : -INFINITY;
    }
    __syncthreads();
    below = bcast_k;
    if (below == 0ull) {  // fewer than top_n elements: the remaining places say so
      if (tid == 0)
        for (int q = r + 1; q < a.top_n; ++q) {
          idx_out[q] = -1;
          val_out[q] = -INFINITY;
        }
      break;
    }
  }
}

}  // namespace dihip

extern "C" {

size_t dihip_logits_processor_workspace_bytes(int M, int N) { return (size_t)std::max(M, 0) * (size_t)std::max(N, 0) * sizeof(int); }

int dihip_logits_processor(void* stream, float* logits, int M, int N, const int64_t* ids, int max_len, const int* cur_len, const int* input_len,
                           const float* repetition_penalty, const float* frequency_penalty, const float* presence_penalty,
                           const int* no_repeat_ngram_size, const int* min_length, const int* eos_token_id,
                           const int* suppress_repetition_in_generation, void* ws, size_t ws_bytes) {
  using namespace dihip;
  DIHIP_REQUIRE(M >= 0 && N > 0 && max_len >= 0 && logits && ids && cur_len && input_len && repetition_penalty && frequency_penalty &&
                    presence_penalty && no_repeat_ngram_size && min_length && eos_token_id && suppress_repetition_in_generation,
                DIHIP_PARAM_ERROR, "logits_processor: bad argument");
  DIHIP_REQUIRE(ws && ws_bytes >= dihip_logits_processor_workspace_bytes(M, N), DIHIP_PARAM_ERROR, "logits_processor: workspace of %zu bytes, %zu needed",
                ws_bytes, dihip_logits_processor_workspace_bytes(M, N));
  DIHIP_REQUIRE(N < (1 << 30), DIHIP_PARAM_ERROR, "logits_processor: vocabulary too large");
  if (M == 0) return DIHIP_SUCCESS;
  LogitsProcArgs a{logits, ids, nullptr, nullptr, max_len, N, cur_len, input_len, repetition_penalty, frequency_penalty, presence_penalty, no_repeat_ngram_size,
                   min_length, eos_token_id, suppress_repetition_in_generation, reinterpret_cast<int*>(ws)};
  hipLaunchKernelGGL(logits_processor_kernel, dim3(M), dim3(LP_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "logits_processor: launch failed: %s", hipGetErrorString(e));
  return DIHIP_SUCCESS;
}

int dihip_logits_processor_rows(void* stream, float* logits, int M, int N, int64_t* const* history_rows, const int64_t* append_ids, int max_len,
                                const int* cur_len, const int* input_len, const float* repetition_penalty, const float* frequency_penalty,
                                const float* presence_penalty, const int* no_repeat_ngram_size, const int* min_length, const int* eos_token_id,
                                const int* suppress_repetition_in_generation, void* ws, size_t ws_bytes) {
  using namespace dihip;
  DIHIP_REQUIRE(M >= 0 && N > 0 && max_len >= 0 && logits && history_rows && cur_len && input_len && repetition_penalty && frequency_penalty &&
                    presence_penalty && no_repeat_ngram_size && min_length && eos_token_id && suppress_repetition_in_generation,
                DIHIP_PARAM_ERROR, "logits_processor_rows: bad argument");
  DIHIP_REQUIRE(ws && ws_bytes >= dihip_logits_processor_workspace_bytes(M, N), DIHIP_PARAM_ERROR, "logits_processor_rows: workspace of %zu bytes, %zu needed",
                ws_bytes, dihip_logits_processor_workspace_bytes(M, N));
  DIHIP_REQUIRE(N < (1 << 30), DIHIP_PARAM_ERROR, "logits_processor_rows: vocabulary too large");
  if (M == 0) return DIHIP_SUCCESS;
  LogitsProcArgs a{logits, nullptr, history_rows, append_ids, max_len, N, cur_len, input_len, repetition_penalty, frequency_penalty, presence_penalty,
                   no_repeat_ngram_size, min_length, eos_token_id, suppress_repetition_in_generation, reinterpret_cast<int*>(ws)};
  hipLaunchKernelGGL(logits_processor_kernel, dim3(M), dim3(LP_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "logits_processor_rows: launch failed: %s", hipGetErrorString(e));
  return DIHIP_SUCCESS;
}

int dihip_logprobs_records(void* stream, const float* logits, int M, int N, const int64_t* chosen, int top_n, int out_stride, float* const* records,
                           const uint32_t* position, int position_bias, int max_records) {
  using namespace dihip;
  DIHIP_REQUIRE(M >= 0 && N > 0 && logits && records && top_n >= 0 && top_n <= LP_MAX_TOP && out_stride >= top_n && out_stride >= 1 && max_records >= 0,
                DIHIP_PARAM_ERROR, "logprobs_records: bad argument");
  if (M == 0) return DIHIP_SUCCESS;
  LogprobArgs a{logits, chosen, N, top_n, out_stride, nullptr, nullptr, nullptr, records, position, position_bias, max_records};
  hipLaunchKernelGGL(logprobs_kernel, dim3(M), dim3(LP_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "logprobs_records: launch failed: %s", hipGetErrorString(e));
  return DIHIP_SUCCESS;
}

int dihip_logprobs(void* stream, const float* logits, int M, int N, const int64_t* chosen, int top_n, int out_stride, float* token_logprob,
                   float* top_value, int* top_index) {
  using namespace dihip;
  DIHIP_REQUIRE(M >= 0 && N > 0 && logits && top_n >= 0 && top_n <= LP_MAX_TOP && out_stride >= top_n, DIHIP_PARAM_ERROR, "logprobs: bad argument");
  DIHIP_REQUIRE(top_n == 0 || (top_value && top_index), DIHIP_PARAM_ERROR, "logprobs: top_n = %d without outputs", top_n);
  if (M == 0) return DIHIP_SUCCESS;
  LogprobArgs a{logits, chosen, N, top_n, out_stride, token_logprob, top_value, top_index, nullptr, nullptr, 0, 0};
  hipLaunchKernelGGL(logprobs_kernel, dim3(M), dim3(LP_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "logprobs: launch failed: %s", hipGetErrorString(e));
  return DIHIP_SUCCESS;
}

}  // extern "C"
