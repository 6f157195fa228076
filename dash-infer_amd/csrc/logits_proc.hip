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
//     log-softmax of the processed logits, the chosen token's value, the top `top_logprobs` <= 10 values and indices): two launches that
//     fill the chip at any batch (a row is split into chunks of 4096 logits held in registers; see logprobs_chunk_kernel); the [batch, vocab]
//     log-probability tensor the reference materialises is never written.
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
      __syncthreads();
    }
    ids = h;
  }
  const int in_len = a.input_len[row];
  const int lo_rep = a.suppress[row] != 0 ? max(in_len, 0) : 0;
  const int lo_gen = max(in_len, 0);
  const float p = a.repetition[row], fq = a.frequency[row], pr = a.presence[row];
  auto valid = [&](int64_t t) { return t >= 0 && t < (int64_t)a.N; };

  // Between the phases: __syncthreads() alone -- its workgroup-scope fence orders this workgroup's stores and atomics (one CU, one L1, the
  // atomics at its L2); a device-scope __threadfence() writes the L2 back at every phase and was half of the kernel's time (24.5 -> 13.8 us).
  // Every sweep takes its positions EIGHT per lane and pass, the eight id loads (and then the eight stores / atomics) in flight together: a
  // lane that walks its positions one by one pays a memory round trip per position and phase (measured: 5 us per 1024 tokens of history).
  constexpr int U = 8;
  // (1) the scratch entries of this request's tokens start at zero (plain stores: no memset of [M, N], no state between calls)
  for (int i0 = tid; i0 < L; i0 += U * LP_THREADS) {
    int64_t t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = i0 + u * LP_THREADS < L ? ids[i0 + u * LP_THREADS] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (valid(t[u])) cnt[t[u]] = 0;
  }
  __syncthreads();
  // (2) count the GENERATED tokens (atomics whose result nobody waits for)
  for (int i0 = lo_gen + tid; i0 < L; i0 += U * LP_THREADS) {
    int64_t t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = i0 + u * LP_THREADS < L ? ids[i0 + u * LP_THREADS] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (valid(t[u])) atomicAdd(&cnt[t[u]], 1);
  }
  __syncthreads();
  // (3) one lane per DISTINCT token of the repetition range claims it (bit 30) and applies, in the reference's order, the repetition
  //     penalty to the unprocessed score and then the frequency / presence penalty of its count.  The generated range lies inside the
  //     repetition range (lo_rep is 0 or input_len), so every counted token is claimed here; a prompt-only token has count 0 and its
  //     `score -= 0` is the identity the reference's full-vocabulary sweep performs
  for (int i0 = lo_rep + tid; i0 < L; i0 += U * LP_THREADS) {
    int64_t t[U];
    int old[U];
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = i0 + u * LP_THREADS < L ? ids[i0 + u * LP_THREADS] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) old[u] = valid(t[u]) ? atomicOr(&cnt[t[u]], 1 << 30) : (1 << 30);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (old[u] >> 30) ? 0.f : s[t[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (old[u] >> 30) continue;
      float w = v[u] < 0.f ? v[u] * p : v[u] / p;
      if (old[u] > 0) {
        float total = (float)old[u] * fq;   // token_count[tid] * frequency_penalty_list[batch]
        total = total + pr;
        w = w - total;
      }
      s[t[u]] = w;
    }
  }
  __syncthreads();
  // (4) no-repeat n-gram (cur_len as given: the comparison window ends at the last token): the first token of the window decides for
  //     almost every position; the rest of the comparison runs for the few that pass
  const int ng = a.ngram[row];
  if (ng > 0) {
    const int64_t tail0 = ng >= 2 && L - ng + 1 >= 0 && L - ng + 1 < L ? ids[L - ng + 1] : 0;
    for (int i0 = tid; i0 < L; i0 += U * LP_THREADS) {
      int64_t t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * LP_THREADS;
        t[u] = (i < L && i + ng - 2 < L - 1) ? ids[i] : -1;  // (-1: not a candidate; a real id -1 never matches a candidate either way)
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * LP_THREADS;
        if (!(i < L && i + ng - 2 < L - 1)) continue;
        bool same = ng < 2 || t[u] == tail0;
        for (int j = 1; j < ng - 1 && same; ++j) same = ids[i + j] == ids[L - ng + j + 1];
        if (same) {
          const int64_t b = ids[i + ng - 1];
          if (valid(b)) s[b] = -1e9f;
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
  int N, top_n, out_stride, chunks;
  float* token_logprob;  // [M] (may be null)
  float* top_value;      // [M, out_stride]
  int* top_index;        // [M, out_stride]
  // records form (top_value == null): row m writes {token_logprob, top values [out_stride], top indices [out_stride]} = 1 + 2 * out_stride
  // words at records[m] + (position[m] + position_bias) * (1 + 2 * out_stride); a null records[m]: the request did not ask
  float* const* records;
  const uint32_t* position;
  int position_bias, max_records;
  unsigned char* ws;     // [M][chunks] partials: {chunk max f32, pad, sum of exp(x - chunk max) f64, top_n keys u64}
};

__device__ __forceinline__ uint32_t lp_order_key(float v) {  // larger float <=> larger key; NaN below everything (as csrc/sample.hip)
  uint32_t b = __float_as_uint(v);
  if ((b & 0x7FFFFFFFu) > 0x7F800000u) return 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// 64-bit sort key: (value key, ~index) -- the larger key is the better candidate (value descending, index ascending); 0 = no element
__device__ __forceinline__ uint64_t lp_pack(float v, int idx) { return ((uint64_t)lp_order_key(v) << 32) | (uint32_t)(~(uint32_t)idx); }

// maximum over the 64 lanes, result in every lane: DPP moves inside the rows of 16 and the two cross-row swaps of device_utils.h's
// wave_max, on both halves of the key -- VALU only (__shfl_xor would be twelve dependent ds_bpermute round trips per reduction, and a
// selection round is little else)
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_mov_u64(uint64_t v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, true);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t max_u64(uint64_t a, uint64_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
  v = max_u64(v, dpp_mov_u64<0xB1>(v));
  v = max_u64(v, dpp_mov_u64<0x4E>(v));
  v = max_u64(v, dpp_mov_u64<0x141>(v));
  v = max_u64(v, dpp_mov_u64<0x140>(v));
  {
    const auto l = __builtin_amdgcn_permlane32_swap((uint32_t)v, (uint32_t)v, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((uint32_t)(v >> 32), (uint32_t)(v >> 32), false, false);
    v = max_u64(((uint64_t)h[0] << 32) | l[0], ((uint64_t)h[1] << 32) | l[1]);
  }
  const auto l = __builtin_amdgcn_permlane16_swap((uint32_t)v, (uint32_t)v, false, false);
  const auto h = __builtin_amdgcn_permlane16_swap((uint32_t)(v >> 32), (uint32_t)(v >> 32), false, false);
  return max_u64(((uint64_t)h[0] << 32) | l[0], ((uint64_t)h[1] << 32) | l[1]);
}

// The log-probabilities in two launches that fill the chip whatever the batch: a row of the Qwen2 vocabulary is 38 chunks of 4096 logits,
//   A  one workgroup per (row, chunk) holds its 4096 logits in registers (16 per lane, coalesced): chunk maximum, sum of exp(x - max) in f64,
//      and the chunk's own top_n keys by top_n block-wide maximum rounds over registers;
//   B  one workgroup per row merges the partials in a fixed order (row maximum, log-sum-exp, top_n rounds over chunks x top_n candidates).
// (The first version walked the row with ONE workgroup, 2 + top_n passes of 600 KB at the ~30 GB/s one workgroup's dependent loads reach:
// 38 us without and 271 us with top-10 at batch 1; this one 8.3 and 18.9 us -- two launches and ~1 us per selection round,
// profiles/r06_logits_proc_timing.txt.)
constexpr int LPC_THREADS = 256, LPC_E = 16, LPC_CHUNK = LPC_THREADS * LPC_E;

__host__ __device__ inline size_t lp_partial_bytes(int top_n) { return 16 + (size_t)top_n * 8; }

__device__ __forceinline__ bool lp_row_skipped(const LogprobArgs& a, int row) {  // records form: the request did not ask / its position is outside
  if (!a.records) return false;
  const long long p = (long long)(a.position ? a.position[row] : 0u) + a.position_bias;
  return !a.records[row] || p < 0 || p >= a.max_records;
}

__global__ __launch_bounds__(LPC_THREADS) void logprobs_chunk_kernel(const LogprobArgs a) {
  __shared__ float red_f[LPC_THREADS / 64];
  __shared__ double red_d[LPC_THREADS / 64];
  __shared__ uint64_t red_k[2][LPC_THREADS / 64];  // (two copies: one barrier per selection round)
  const int chunk = blockIdx.x, row = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (lp_row_skipped(a, row)) return;  // (uniform over the workgroup)
  const float* x = a.logits + (size_t)row * a.N;
  const int base = chunk * LPC_CHUNK + tid;
  float v[LPC_E];
  uint64_t key[LPC_E];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < LPC_E; ++j) {
    const int i = base + j * LPC_THREADS;
    v[j] = i < a.N ? x[i] : -INFINITY;
    key[j] = i < a.N ? lp_pack(v[j], i) : 0ull;
    mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx);
  if (lane == 0) red_f[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
  double sum = 0.0;
  if (mx > -INFINITY) {
#pragma unroll
    for (int j = 0; j < LPC_E; ++j) sum += (double)expf(v[j] - mx);  // (exp(-inf) = 0 for the places past the row)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) red_d[wave] = sum;
  __syncthreads();
  unsigned char* part = a.ws + ((size_t)row * a.chunks + chunk) * lp_partial_bytes(a.top_n);
  if (tid == 0) {
    *reinterpret_cast<float*>(part) = mx;
    *reinterpret_cast<double*>(part + 8) = ((red_d[0] + red_d[1]) + red_d[2]) + red_d[3];
  }
  uint64_t* keys_out = reinterpret_cast<uint64_t*>(part + 16);
  uint64_t below = ~0ull;
  for (int r = 0; r < a.top_n; ++r) {
    uint64_t best = 0ull;
#pragma unroll
    for (int j = 0; j < LPC_E; ++j) best = (key[j] < below && key[j] > best) ? key[j] : best;
    best = wave_max_u64(best);
    if (lane == 0) red_k[r & 1][wave] = best;
    __syncthreads();
    uint64_t b = red_k[r & 1][0];
#pragma unroll
    for (int w = 1; w < LPC_THREADS / 64; ++w) b = max_u64(b, red_k[r & 1][w]);
    if (tid == 0) keys_out[r] = b;
    below = b;  // 0: the chunk is exhausted -- the remaining places stay 0 as well (nothing is < 0)
  }
}

__global__ __launch_bounds__(LPC_THREADS) void logprobs_merge_kernel(const LogprobArgs a) {
  __shared__ float red_f[LPC_THREADS / 64];
  __shared__ double red_d[LPC_THREADS / 64];
  __shared__ uint64_t red_k[2][LPC_THREADS / 64];  // (two copies: one barrier per selection round)
  __shared__ uint64_t sel[LP_MAX_TOP];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (lp_row_skipped(a, row)) return;
  const float* x = a.logits + (size_t)row * a.N;
  float* tok_out = a.token_logprob ? a.token_logprob + row : nullptr;
  float* val_out = a.top_value ? a.top_value + (size_t)row * a.out_stride : nullptr;
  int* idx_out = a.top_index ? a.top_index + (size_t)row * a.out_stride : nullptr;
  if (a.records) {
    float* rec = a.records[row] + (size_t)((long long)(a.position ? a.position[row] : 0u) + a.position_bias) * (1 + 2 * a.out_stride);
    tok_out = rec;
    val_out = rec + 1;
    idx_out = reinterpret_cast<int*>(rec + 1 + a.out_stride);
  }
  const size_t pb = lp_partial_bytes(a.top_n);
  const unsigned char* parts = a.ws + (size_t)row * a.chunks * pb;
  // row maximum over the chunk maxima, then the chunk sums rescaled to it -- thread-strided, wave butterfly, waves in order: a fixed order
  float mx = -INFINITY;
  for (int c = tid; c < a.chunks; c += LPC_THREADS) mx = fmaxf(mx, *reinterpret_cast<const float*>(parts + (size_t)c * pb));
  mx = wave_max(mx);
  if (lane == 0) red_f[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
  double sum = 0.0;
  for (int c = tid; c < a.chunks; c += LPC_THREADS) {
    const float mc = *reinterpret_cast<const float*>(parts + (size_t)c * pb);
    const double sc = *reinterpret_cast<const double*>(parts + (size_t)c * pb + 8);
    if (mc > -INFINITY) sum += sc * exp((double)mc - (double)mx);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) red_d[wave] = sum;
  __syncthreads();
  const float lse = (float)log(((red_d[0] + red_d[1]) + red_d[2]) + red_d[3]);
  if (tid == 0 && tok_out && a.chosen) {
    const int64_t c = a.chosen[row];
    *tok_out = (c >= 0 && c < (int64_t)a.N) ? (x[c] - mx) - lse : -INFINITY;
  }
  // top_n of the chunks' candidates (chunks x top_n keys; a chunk's r-th key can only win after its (r-1)-th: all are candidates)
  const int ncand = a.chunks * a.top_n;
  // the candidates sit in registers for the rounds (up to 8 per lane: 2048 -- the Qwen2 vocabulary with top-10 has 380); beyond, re-read
  constexpr int CR = 8;
  const bool in_regs = ncand <= CR * LPC_THREADS;
  auto cand = [&](int i) { return *reinterpret_cast<const uint64_t*>(parts + (size_t)(i / a.top_n) * pb + 16 + (size_t)(i % a.top_n) * 8); };
  uint64_t ck[CR];
#pragma unroll
  for (int j = 0; j < CR; ++j) ck[j] = (in_regs && tid + j * LPC_THREADS < ncand) ? cand(tid + j * LPC_THREADS) : 0ull;
  uint64_t below = ~0ull;
  for (int r = 0; r < a.top_n; ++r) {
    uint64_t best = 0ull;
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < CR; ++j) best = (ck[j] < below && ck[j] > best) ? ck[j] : best;
    } else {
      for (int i = tid; i < ncand; i += LPC_THREADS) {
        const uint64_t k = cand(i);
        best = (k < below && k > best) ? k : best;
      }
    }
    best = wave_max_u64(best);
    if (lane == 0) red_k[r & 1][wave] = best;
    __syncthreads();
    uint64_t b = red_k[r & 1][0];
#pragma unroll
    for (int w = 1; w < LPC_THREADS / 64; ++w) b = max_u64(b, red_k[r & 1][w]);
    if (tid == 0) sel[r] = b;
    below = b;
  }
  __syncthreads();
  // lane r writes place r: the value comes back out of the key (no dependent load inside the rounds)
  if (tid < a.top_n) {
    const uint64_t b = sel[tid];
    const int idx = (int)(~(uint32_t)(b & 0xFFFFFFFFu));
    const bool any = b != 0ull && idx >= 0 && idx < a.N;
    const uint32_t k = (uint32_t)(b >> 32);
    const float v = __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);  // inverse of lp_order_key
    idx_out[tid] = any ? idx : -1;                               // (places beyond the row's length say so)
    val_out[tid] = any ? (v - mx) - lse : -INFINITY;
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

size_t dihip_logprobs_workspace_bytes(int M, int N, int top_n) {
  const size_t chunks = ((size_t)std::max(N, 0) + dihip::LPC_CHUNK - 1) / dihip::LPC_CHUNK;
  return (size_t)std::max(M, 0) * chunks * dihip::lp_partial_bytes(std::max(top_n, 0));
}

static int logprobs_launch(void* stream, dihip::LogprobArgs a, int M, void* ws, size_t ws_bytes, const char* what) {
  using namespace dihip;
  DIHIP_REQUIRE(ws && (reinterpret_cast<uintptr_t>(ws) & 7) == 0 && ws_bytes >= dihip_logprobs_workspace_bytes(M, a.N, a.top_n), DIHIP_PARAM_ERROR,
                "%s: workspace of %zu bytes (8-byte aligned), %zu needed", what, ws_bytes, dihip_logprobs_workspace_bytes(M, a.N, a.top_n));
  a.chunks = (a.N + LPC_CHUNK - 1) / LPC_CHUNK;
  a.ws = reinterpret_cast<unsigned char*>(ws);
  DIHIP_REQUIRE(M <= 65535, DIHIP_PARAM_ERROR, "%s: more than 65535 rows", what);
  hipLaunchKernelGGL(logprobs_chunk_kernel, dim3(a.chunks, M), dim3(LPC_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  hipLaunchKernelGGL(logprobs_merge_kernel, dim3(M), dim3(LPC_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "%s: launch failed: %s", what, hipGetErrorString(e));
  return DIHIP_SUCCESS;
}

int dihip_logprobs_records(void* stream, const float* logits, int M, int N, const int64_t* chosen, int top_n, int out_stride, float* const* records,
                           const uint32_t* position, int position_bias, int max_records, void* ws, size_t ws_bytes) {
  using namespace dihip;
  DIHIP_REQUIRE(M >= 0 && N > 0 && logits && records && top_n >= 0 && top_n <= LP_MAX_TOP && out_stride >= top_n && out_stride >= 1 && max_records >= 0,
                DIHIP_PARAM_ERROR, "logprobs_records: bad argument");
  if (M == 0) return DIHIP_SUCCESS;
  LogprobArgs a{logits, chosen, N, top_n, out_stride, 0, nullptr, nullptr, nullptr, records, position, position_bias, max_records, nullptr};
  return logprobs_launch(stream, a, M, ws, ws_bytes, "logprobs_records");
}

int dihip_logprobs(void* stream, const float* logits, int M, int N, const int64_t* chosen, int top_n, int out_stride, float* token_logprob,
                   float* top_value, int* top_index, void* ws, size_t ws_bytes) {
  using namespace dihip;
  DIHIP_REQUIRE(M >= 0 && N > 0 && logits && top_n >= 0 && top_n <= LP_MAX_TOP && out_stride >= top_n, DIHIP_PARAM_ERROR, "logprobs: bad argument");
  DIHIP_REQUIRE(top_n == 0 || (top_value && top_index), DIHIP_PARAM_ERROR, "logprobs: top_n = %d without outputs", top_n);
  if (M == 0) return DIHIP_SUCCESS;
  LogprobArgs a{logits, chosen, N, top_n, out_stride, 0, token_logprob, top_value, top_index, nullptr, nullptr, 0, 0, nullptr};
  return logprobs_launch(stream, a, M, ws, ws_bytes, "logprobs");
}

}  // extern "C"
