// sample.hip -- top-k / top-p / temperature sampling over f32 logits: the sampling half of GenerateOp
// (csrc/core/operator/generate_opt/generate/generate_op.cpp:472-600 RunSample; the arithmetic of its x86 path,
// generate_impl_cpu.hpp:120-170 gen_sample_cpu):
//     TopK (the k largest logits, descending)  ->  softmax(logit / T) over them  ->  TopP (kernel/cpu/topp.cpp:14-30: the shortest
//     prefix whose cumulated probability EXCEEDS p; skipped for p <= 1e-7)  ->  softmax(logit / T) over that prefix  ->  the
//     exponential race of kernel/cpu/sample.cpp:42-68: score_i = prob_i / q_i, q_i = -log1p(-u_i), u_i uniform in [0, 1), the first
//     maximum wins.
// top_k == 0 means "the whole vocabulary" in the reference (real_k = vocab_size_, generate_op.cpp:338-339: pure top-p sampling), and its
// default build serves any k (CONFIG_SAMPLE_CONSTRAIN_MAX_K, which refuses k > 1024, is not defined anywhere in it).  Rows with
// 1 <= k <= 1024 run sample_kernel below (candidates sorted in LDS); rows with k == 0 or k > 1024 run sample_wide_kernel (round 6,
// ADVICE r5): the same pipeline WITHOUT materialising the sorted candidate list -- see its header.  top_k == 1 is greedy: the arg-max,
// lowest index on ties.
// The random stream is this backend's own (the reference draws from std::mt19937 on x86 and Philox on CUDA: neither is
// reproducible on another device): u_i = 24 high bits of splitmix64(seed, position of the sampled token, candidate rank i) -- a
// pure function of (request seed, sequence position, rank), so a captured decode step replays correctly with the positions read
// from the device.  oracle/sampling.py restates the whole pipeline in numpy with the same stream.
//
// One workgroup of 1024 threads per row: radix select of the k-th largest key (4 passes of 8 bits over an order-preserving
// integer image of the floats, 256-bin LDS histograms), ordered compaction (ties at the threshold in index order), bitonic sort
// of the <= 1024 candidates in LDS by (value descending, index ascending), then the softmax / top-p / race with block scans.
#include "device_utils.h"
#include "dashinfer_hip.h"

namespace dihip {

constexpr int SAMPLE_THREADS = 1024;
constexpr int SAMPLE_MAX_K = 1024;

__device__ __forceinline__ uint32_t order_key(float v) {  // larger float <=> larger unsigned key; NaN sorts below everything
  uint32_t b = __float_as_uint(v);
  if ((b & 0x7FFFFFFFu) > 0x7F800000u) return 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float uniform01(uint64_t seed, uint32_t position, uint32_t rank) {
  const uint64_t h = splitmix64(splitmix64(seed ^ 0xD1B54A32D192ED03ull) + (((uint64_t)position << 32) | rank));
  return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);  // 24 bits: [0, 1)
}

// block-wide inclusive scan (sum) of one float per thread; `tmp`: 32 floats of LDS
__device__ __forceinline__ float block_scan_sum(float v, float* tmp, float* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  __syncthreads();
  if (lane == 63) tmp[wave] = v;
  __syncthreads();
  float base = 0.f, tot = 0.f;
  for (int w = 0; w < SAMPLE_THREADS / 64; ++w) {
    const float t = tmp[w];
    if (w < wave) base += t;
    tot += t;
  }
  if (total) *total = tot;
  return v + base;
}
__device__ __forceinline__ unsigned block_scan_u32(unsigned v, unsigned* tmp, unsigned* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  __syncthreads();
  if (lane == 63) tmp[wave] = v;
  __syncthreads();
  unsigned base = 0, tot = 0;
  for (int w = 0; w < SAMPLE_THREADS / 64; ++w) {
    const unsigned t = tmp[w];
    if (w < wave) base += t;
    tot += t;
  }
  if (total) *total = tot;
  return v + base;
}

struct SampleArgs {
  int64_t* ids;           // [M]
  const float* logits;    // [M, N]
  int N;
  const int* top_k;       // [M]
  const float* top_p;     // [M]
  const float* temperature;  // [M]
  const uint64_t* seed;   // [M]
  const uint32_t* position;  // [M] position of the token being sampled (device-resident sequence length), or null: 0
  uint32_t* counters_a;   // advanced by one per row after sampling (the decode step's length counters), or null
  uint32_t* counters_b;
  float* probs_out;       // diagnostics / tests: [M, 1024] final probabilities of the sorted candidates (0 beyond), or null
  int* cand_out;          // diagnostics / tests: [M, 1024] candidate indices in sorted order (-1 beyond), or null
};

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_kernel(const SampleArgs a) {
  __shared__ unsigned hist[256];
  __shared__ float cval[SAMPLE_MAX_K];
  __shared__ int cidx[SAMPLE_MAX_K];
  __shared__ unsigned su[40];
  __shared__ float sf[40];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = a.logits + (size_t)row * a.N;
  int k = a.top_k[row];
  if (k <= 0 || k > SAMPLE_MAX_K) return;  // a wide row: sample_wide_kernel's (same launch pair, dihip_sample)
  if (k > a.N) k = a.N;
  // contiguous index ranges per thread: thread order == index order (ordered compaction of the ties)
  const int per = (a.N + SAMPLE_THREADS - 1) / SAMPLE_THREADS;
  const int i0 = min(tid * per, a.N), i1 = min(i0 + per, a.N);

  // ---- radix select: the key of the k-th largest element -----------------------------------------------------------------
  uint32_t prefix = 0, pmask = 0;
  unsigned remaining = (unsigned)k;  // rank (1-based, from the top) still to be located inside the current prefix class
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const int shift = pass * 8;
    for (int i = i0; i < i1; ++i) {
      const uint32_t key = order_key(x[i]);
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    if (tid == 0) {  // walk the digits from the top: the bin holding the `remaining`-th largest
      unsigned acc = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (acc + hist[d] >= remaining) break;
        acc += hist[d];
      }
      su[32] = (unsigned)d;
      su[33] = remaining - acc;
    }
    __syncthreads();
    prefix |= su[32] << shift;
    pmask |= 0xFFu << shift;
    remaining = su[33];
    __syncthreads();
  }
  const uint32_t thr = prefix;         // key of the k-th largest element
  const unsigned ties_wanted = remaining;  // how many elements EQUAL to it belong to the top k (lowest indices first)

  // ---- ordered compaction ---------------------------------------------------------------------------------------------------
  unsigned ngt = 0, neq = 0;
  for (int i = i0; i < i1; ++i) {
    const uint32_t key = order_key(x[i]);
    ngt += key > thr;
    neq += key == thr;
  }
  unsigned tot_gt = 0;
  const unsigned gt_end = block_scan_u32(ngt, su, &tot_gt);
  __syncthreads();
  const unsigned eq_end = block_scan_u32(neq, su, nullptr);
  unsigned gpos = gt_end - ngt, epos = eq_end - neq;
  for (int i = tid; i < SAMPLE_MAX_K; i += SAMPLE_THREADS) {
    cval[i] = -INFINITY;
    cidx[i] = 0x7FFFFFFF;
  }
  __syncthreads();
  for (int i = i0; i < i1; ++i) {
    const float v = x[i];
    const uint32_t key = order_key(v);
    if (key > thr) {
      cval[gpos] = v;
      cidx[gpos] = i;
      ++gpos;
    } else if (key == thr) {
      if (epos < ties_wanted) {
        cval[tot_gt + epos] = v;
        cidx[tot_gt + epos] = i;
      }
      ++epos;
    }
  }
  __syncthreads();

  // ---- bitonic sort of the 1024 slots: value descending, index ascending (padding: -inf, max index -> the tail) -------------
  for (int size = 2; size <= SAMPLE_MAX_K; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int j = tid ^ stride;
      if (j > tid) {
        const bool desc = (tid & size) == 0;
        const float va = cval[tid], vb = cval[j];
        const int ia = cidx[tid], ib = cidx[j];
        const bool a_first = va > vb || (va == vb && ia < ib);  // a belongs before b in the final order
        if (desc ? !a_first : a_first) {
          cval[tid] = vb;
          cval[j] = va;
          cidx[tid] = ib;
          cidx[j] = ia;
        }
      }
      __syncthreads();
    }
  }

  // ---- softmax(T) -> top-p cut -> softmax(T) over the prefix -> exponential race -----------------------------------------------
  const float T = a.temperature[row];
  const float inv_t = 1.0f / T;
  const float vmax = cval[0];
  const bool mine = tid < k;
  const float e = mine ? expf((cval[tid] - vmax) * inv_t) : 0.f;
  float sum1 = 0.f;
  const float cum = block_scan_sum(e, sf, &sum1);
  __syncthreads();
  int kk = k;
  const float p = a.top_p[row];
  if (p > 1e-7f) {
    // the first rank whose cumulated probability exceeds p closes the prefix (topp.cpp:20-26); none: all k stay
    if (tid == 0) su[34] = (unsigned)k;
    __syncthreads();
    if (mine && cum / sum1 > p) atomicMin(&su[34], (unsigned)(tid + 1));
    __syncthreads();
    kk = (int)su[34];
  }
  const bool in2 = tid < kk;
  float sum2 = 0.f;
  (void)block_scan_sum(in2 ? e : 0.f, sf, &sum2);
  __syncthreads();
  const float prob = in2 ? e / sum2 : 0.f;
  if (a.probs_out) a.probs_out[(size_t)row * SAMPLE_MAX_K + tid] = prob;
  if (a.cand_out) a.cand_out[(size_t)row * SAMPLE_MAX_K + tid] = tid < k ? cidx[tid] : -1;
  const uint32_t pos = a.position ? a.position[row] : 0u;
  float score = -1.f;
  if (in2) {
    const float u = uniform01(a.seed[row], pos, (uint32_t)tid);
    const float q = -log1pf(-u);          // Exp(1); u == 0 gives q == 0 and an infinite score: that rank wins, as in the reference
    score = prob / q;
  }
  // arg-max of the scores, first maximum (lowest rank) wins
  float bs = score;
  int br = tid;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float os = __shfl_xor(bs, off, 64);
    const int orr = __shfl_xor(br, off, 64);
    if (os > bs || (os == bs && orr < br)) {
      bs = os;
      br = orr;
    }
  }
  __syncthreads();
  if ((tid & 63) == 0) {
    sf[tid >> 6] = bs;
    su[tid >> 6] = (unsigned)br;
  }
  __syncthreads();
  if (tid == 0) {
    float best = sf[0];
    unsigned rank = su[0];
    for (int w = 1; w < SAMPLE_THREADS / 64; ++w)
      if (sf[w] > best || (sf[w] == best && su[w] < rank)) {
        best = sf[w];
        rank = su[w];
      }
    a.ids[row] = (int64_t)cidx[rank];
    if (a.counters_a) a.counters_a[row] += 1u;
    if (a.counters_b) a.counters_b[row] += 1u;
  }
}

// ---- wide rows: top_k == 0 (the whole vocabulary) or top_k > 1024 -------------------------------------------------------------------
// The pipeline of sample_kernel -- TopK, softmax(T), TopP prefix, softmax, exponential race -- over up to N candidates, sort-free:
//   * the top-k set is {key > thr_k} + the `ties_k` lowest-index elements with key == thr_k (radix select by COUNT, as above);
//   * e_i = expf((x_i - max) / T); masses are summed as FIXED-POINT integers E_i = (u64)(e_i * 2^32): integer sums are exact and
//     order-independent, so the result does not depend on the order of the LDS atomics (and oracle/sampling.py restates it bit for bit);
//   * the top-p prefix -- the shortest prefix of the (value descending, index ascending) order whose cumulated mass EXCEEDS
//     target = floor(p * sum) -- is found by a radix select by MASS: per digit a 256-bin histogram of mass, walked from the top until the
//     cumulated mass exceeds the target; ties at the threshold key all carry the same mass, so the count of them that belongs to the
//     prefix is floor(remaining / E_tie) + 1;
//   * the final set is the shorter of the two prefixes; the second softmax only scales every candidate's probability by the same 1 / sum2, which
//     does not change the winner of the race, so score_i = e_i / q_i, q_i = -log1p(-u_i);
//   * u_i is keyed by the TOKEN INDEX, u_i = uniform01(seed, position, 0x40000000 + i) (a candidate's rank does not exist without the sort;
//     the stream is this backend's own either way); the first maximum in index order wins.
// One workgroup of 1024 threads per row, ~12 passes over the row (L2-resident: 600 KB at the 7B vocabulary).
__device__ __forceinline__ uint64_t fixed_mass(float e) { return (uint64_t)((double)e * 4294967296.0); }

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_wide_kernel(const SampleArgs a) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long mhist[256];
  __shared__ unsigned su[40];
  __shared__ float sf[40];
  __shared__ unsigned long long sl[40];
  const int row = blockIdx.x, tid = threadIdx.x;
  int k = a.top_k[row];
  if (k >= 1 && k <= SAMPLE_MAX_K) return;  // a narrow row: sample_kernel's
  const float* x = a.logits + (size_t)row * a.N;
  if (k <= 0 || k > a.N) k = a.N;
  const int per = (a.N + SAMPLE_THREADS - 1) / SAMPLE_THREADS;
  const int i0 = min(tid * per, a.N), i1 = min(i0 + per, a.N);
  const int lane = tid & 63, wave = tid >> 6;

  // ---- row maximum (NaN sorts below everything: order_key) ----
  uint32_t kmax = 0;
  for (int i = i0; i < i1; ++i) kmax = max(kmax, order_key(x[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
  if (lane == 0) su[wave] = kmax;
  __syncthreads();
  kmax = 0;
  for (int w = 0; w < SAMPLE_THREADS / 64; ++w) kmax = max(kmax, su[w]);
  __syncthreads();
  // the float of the maximal key (inverse of order_key; all-NaN rows: 0)
  const float vmax = kmax == 0u ? 0.f : __uint_as_float((kmax & 0x80000000u) ? (kmax & 0x7FFFFFFFu) : ~kmax);
  const float inv_t = 1.0f / a.temperature[row];

  // ---- top-k threshold by count (k < N), as sample_kernel ----
  uint32_t thr_k = 0;
  unsigned ties_k = 0xFFFFFFFFu;  // k == N: every element, every tie
  if (k < a.N) {
    uint32_t prefix = 0, pmask = 0;
    unsigned remaining = (unsigned)k;
    for (int pass = 3; pass >= 0; --pass) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const int shift = pass * 8;
      for (int i = i0; i < i1; ++i) {
        const uint32_t key = order_key(x[i]);
        if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned acc = 0;
        int d = 255;
        for (; d > 0; --d) {
          if (acc + hist[d] >= remaining) break;
          acc += hist[d];
        }
        su[32] = (unsigned)d;
        su[33] = remaining - acc;
      }
      __syncthreads();
      prefix |= su[32] << shift;
      pmask |= 0xFFu << shift;
      remaining = su[33];
      __syncthreads();
    }
    thr_k = prefix;
    ties_k = remaining;
  }
  // ties at thr_k in lower-index threads (the ordered part of the membership test)
  unsigned neq = 0;
  if (k < a.N)
    for (int i = i0; i < i1; ++i) neq += order_key(x[i]) == thr_k;
  const unsigned eqk_base = block_scan_u32(neq, su, nullptr) - neq;
  __syncthreads();
  // element i (walked in index order by its thread, `seen` = ties at thr_k met so far) belongs to the top-k set
  auto in_topk = [&](uint32_t key, unsigned& seen) {
    if (key > thr_k) return true;
    if (key < thr_k) return false;
    return seen++ < ties_k;
  };

  // ---- sum of the masses of the top-k set ----
  unsigned long long part = 0;
  {
    unsigned seen = eqk_base;
    for (int i = i0; i < i1; ++i) {
      const float v = x[i];
      if (in_topk(order_key(v), seen)) part += fixed_mass(expf((v - vmax) * inv_t));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += (unsigned long long)__shfl_xor((long long)part, off, 64);
  if (lane == 0) sl[wave] = part;
  __syncthreads();
  unsigned long long sum1 = 0;
  for (int w = 0; w < SAMPLE_THREADS / 64; ++w) sum1 += sl[w];
  __syncthreads();

  // ---- top-p prefix by mass ----
  uint32_t thr = thr_k;
  unsigned ties_wanted = ties_k;
  const float p = a.top_p[row];
  if (p > 1e-7f) {
    const unsigned long long target = (unsigned long long)((double)p * (double)sum1);  // the prefix's mass must EXCEED this
    if (sum1 > target) {  // (else no prefix exceeds p: all of the top-k set stays)
      uint32_t prefix = 0, pmask = 0;
      unsigned long long remaining = target;
      for (int pass = 3; pass >= 0; --pass) {
        if (tid < 256) mhist[tid] = 0ull;
        __syncthreads();
        const int shift = pass * 8;
        unsigned seen = eqk_base;
        for (int i = i0; i < i1; ++i) {
          const float v = x[i];
          const uint32_t key = order_key(v);
          if (!in_topk(key, seen)) continue;
          if ((key & pmask) == prefix) atomicAdd(&mhist[(key >> shift) & 0xFF], fixed_mass(expf((v - vmax) * inv_t)));
        }
        __syncthreads();
        if (tid == 0) {  // the digit in which the cumulated mass first exceeds what is still to be covered
          unsigned long long acc = 0;
          int d = 255;
          for (; d > 0; --d) {
            if (acc + mhist[d] > remaining) break;
            acc += mhist[d];
          }
          su[32] = (unsigned)d;
          sl[32] = remaining - acc;
        }
        __syncthreads();
        prefix |= su[32] << shift;
        pmask |= 0xFFu << shift;
        remaining = sl[32];
        __syncthreads();
      }
      // all top-k elements with key == prefix carry one mass E: the smallest count t with t * E > remaining
      const float vt = __uint_as_float((prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix);
      const unsigned long long E = fixed_mass(expf((vt - vmax) * inv_t));
      const unsigned long long t = E ? remaining / E + 1ull : 0xFFFFFFFFull;
      const unsigned tp = t > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)t;
      // the shorter prefix: a larger threshold key is stricter; equal keys: fewer ties
      if (prefix > thr_k || k >= a.N) {
        thr = prefix;
        ties_wanted = tp;
      } else {  // prefix == thr_k (it cannot be smaller: only top-k elements carry mass)
        thr = thr_k;
        ties_wanted = min(ties_k, tp);
      }
    }
  }

  // ---- the race over the final set: score = e / q, the first maximum in index order wins ----
  unsigned neq2 = 0;
  for (int i = i0; i < i1; ++i) neq2 += order_key(x[i]) == thr;
  const unsigned eq_base = block_scan_u32(neq2, su, nullptr) - neq2;
  __syncthreads();
  const uint32_t pos = a.position ? a.position[row] : 0u;
  const uint64_t seed = a.seed[row];
  float bs = -1.f;
  int bi = 0x7FFFFFFF;
  {
    unsigned seen = eq_base;
    for (int i = i0; i < i1; ++i) {
      const float v = x[i];
      const uint32_t key = order_key(v);
      bool in = key > thr;
      if (key == thr) in = seen++ < ties_wanted;
      if (!in) continue;
      const float e = expf((v - vmax) * inv_t);
      const float u = uniform01(seed, pos, 0x40000000u + (uint32_t)i);
      const float q = -log1pf(-u);
      const float sc = e / q;  // u == 0: infinite score, that candidate wins (as in the reference's race)
      if (sc > bs) {           // (index order inside a thread: the first maximum stays)
        bs = sc;
        bi = i;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float os = __shfl_xor(bs, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (os > bs || (os == bs && oi < bi)) {
      bs = os;
      bi = oi;
    }
  }
  if (lane == 0) {
    sf[wave] = bs;
    su[wave] = (unsigned)bi;
  }
  __syncthreads();
  if (tid == 0) {
    float best = sf[0];
    unsigned idx = su[0];
    for (int w = 1; w < SAMPLE_THREADS / 64; ++w)
      if (sf[w] > best || (sf[w] == best && su[w] < idx)) {
        best = sf[w];
        idx = su[w];
      }
    a.ids[row] = (int64_t)(idx == 0x7FFFFFFFu ? 0u : idx);
    if (a.counters_a) a.counters_a[row] += 1u;
    if (a.counters_b) a.counters_b[row] += 1u;
    if (a.cand_out) {  // diagnostics: the final set as (threshold key, ties at it), the fixed-point sum of the top-k set
      a.cand_out[(size_t)row * SAMPLE_MAX_K + 0] = (int)thr;
      a.cand_out[(size_t)row * SAMPLE_MAX_K + 1] = (int)ties_wanted;
      a.cand_out[(size_t)row * SAMPLE_MAX_K + 2] = (int)(sum1 & 0xFFFFFFFFull);
      a.cand_out[(size_t)row * SAMPLE_MAX_K + 3] = (int)(sum1 >> 32);
    }
  }
}

}  // namespace dihip

extern "C" {

static int sample_impl(void* stream, int64_t* ids, const float* logits, int M, int N, const int* top_k, const float* top_p, const float* temperature,
                       const unsigned long long* seed, const uint32_t* position, uint32_t* counters_a, uint32_t* counters_b, float* probs_out,
                       int* cand_out, int wide_rows) {
  using namespace dihip;
  DIHIP_REQUIRE(M >= 0 && N > 0 && ids && logits && top_k && top_p && temperature && seed, DIHIP_PARAM_ERROR, "sample: bad argument");
  if (M == 0) return DIHIP_SUCCESS;
  SampleArgs a{ids, logits, N, top_k, top_p, temperature, reinterpret_cast<const uint64_t*>(seed), position, counters_a, counters_b, probs_out, cand_out};
  // both kernels over all rows: each takes the rows of its kind (the per-row top_k lives on the device; a captured step replays whatever
  // the requests' parameters are).  wide_rows: what the caller knows on the host -- 0 skips the second launch
  hipLaunchKernelGGL(sample_kernel, dim3(M), dim3(SAMPLE_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  if (wide_rows != 0) hipLaunchKernelGGL(sample_wide_kernel, dim3(M), dim3(SAMPLE_THREADS), 0, reinterpret_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "sample: launch failed: %s", hipGetErrorString(e));
  return DIHIP_SUCCESS;
}

int dihip_sample(void* stream, int64_t* ids, const float* logits, int M, int N, const int* top_k, const float* top_p, const float* temperature,
                 const unsigned long long* seed, const uint32_t* position, uint32_t* counters_a, uint32_t* counters_b, float* probs_out,
                 int* cand_out) {
  return sample_impl(stream, ids, logits, M, N, top_k, top_p, temperature, seed, position, counters_a, counters_b, probs_out, cand_out, -1);
}

// wide_rows: how many rows have top_k == 0 or > 1024 (the caller staged the parameters from the host): 0 skips the wide kernel's launch
int dihip_sample_rows(void* stream, int64_t* ids, const float* logits, int M, int N, const int* top_k, const float* top_p, const float* temperature,
                      const unsigned long long* seed, const uint32_t* position, uint32_t* counters_a, uint32_t* counters_b, int wide_rows) {
  return sample_impl(stream, ids, logits, M, N, top_k, top_p, temperature, seed, position, counters_a, counters_b, nullptr, nullptr, wide_rows);
}

}  // extern "C"
