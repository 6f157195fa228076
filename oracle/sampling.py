"""Sampling oracle in numpy (TEST INFRASTRUCTURE ONLY): the x86 path of GenerateOp's sampling half,
csrc/core/operator/generate_opt/generate/generate_impl_cpu.hpp:120-170 (gen_sample_cpu):
    cpu::TopKKernel (k largest logits, descending)
    -> cpu::SoftmaxKernel(values, k, temperature)          probabilities of the k candidates
    -> cpu::TopPKernel (kernel/cpu/topp.cpp:14-30)          k <- 1 + first rank whose cumulated probability EXCEEDS p (p > 1e-7)
    -> cpu::SoftmaxKernel over the first k candidates again
    -> cpu::SampleKernel (kernel/cpu/sample.cpp:42-68)      score_i = prob_i / q_i, q_i = -log1p(-u_i); the first maximum wins
PARITY UNPINNED for the random stream: the reference draws u from std::mt19937 (x86) or Philox (CUDA), which no other device
reproduces; the HIP backend defines its own counter-based stream (csrc/sample.hip: 24 high bits of a splitmix64 hash of
(seed, position, rank)), restated here bit for bit.  top_k == 0 (the whole vocabulary, generate_op.cpp:338-339) or > 1024: sample_wide() below,
the sort-free form of the same pipeline that csrc/sample.hip's sample_wide_kernel runs (fixed-point masses, stream keyed by token index)."""
import numpy as np

MASK = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & MASK
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK
    return x ^ (x >> 31)


def uniform01(seed, position, rank):
    h = _splitmix64((_splitmix64((seed ^ 0xD1B54A32D192ED03) & MASK) + ((position << 32) | rank)) & MASK)
    return np.float32(h >> 40) * np.float32(1.0 / 16777216.0)


def candidates(logits, top_k):
    """-> (indices, values) of the top-k logits: value descending, index ascending on ties (the kernel's documented order)."""
    x = np.asarray(logits, np.float32)
    k = top_k if top_k > 0 else x.shape[0]
    k = min(k, x.shape[0])
    order = np.lexsort((np.arange(x.shape[0]), -x.astype(np.float64)))[:k]
    return order, x[order]


def final_probs(logits, top_k, top_p, temperature):
    idx, v = candidates(logits, top_k)
    e = np.exp((v.astype(np.float64) - float(v[0])) / float(temperature))
    p1 = e / e.sum()
    kk = len(v)
    if top_p > 1e-7:
        cum = np.cumsum(p1)
        over = np.nonzero(cum > top_p)[0]
        if len(over):
            kk = int(over[0]) + 1
    p2 = e[:kk] / e[:kk].sum()
    return idx[:kk], p2


def sample(logits, top_k, top_p, temperature, seed, position=0):
    """-> (token id, margin): margin = best score / second-best score (a comparison with another implementation of exp / log1p is
    only meaningful when it is not ~1)."""
    idx, p = final_probs(logits, top_k, top_p, temperature)
    u = np.array([uniform01(seed, position, r) for r in range(len(idx))], np.float32)
    with np.errstate(divide="ignore"):
        q = -np.log1p(-u.astype(np.float64))
        score = p / q
    best = int(np.argmax(score))           # first maximum
    rest = np.delete(score, best)
    margin = float(score[best] / rest.max()) if len(rest) and rest.max() > 0 else np.inf
    return int(idx[best]), margin


# ---- wide rows (top_k == 0 or > 1024): csrc/sample.hip sample_wide_kernel, restated ---------------------------------------------------
def _fixed_mass(e):
    """(u64)(e * 2^32) of float32 e in (0, 1]"""
    return (np.asarray(e, np.float32).astype(np.float64) * 4294967296.0).astype(np.uint64)


def _expf32(v, vmax, temperature):
    """expf((v - vmax) * (1 / T)) in float32 (the kernel's arithmetic; libm's float32 exp may differ from the device's by an ulp:
    compare the SET and the winner with a margin, not the bits of e)"""
    inv_t = np.float32(1.0) / np.float32(temperature)
    return np.exp(((np.asarray(v, np.float32) - np.float32(vmax)) * inv_t).astype(np.float32)).astype(np.float32)


def wide_final_set(logits, top_k, top_p, temperature):
    """-> (indices of the final candidate set in (value descending, index ascending) order, their float32 e).  The set is a PREFIX of that
    order: the top-k prefix cut again where the cumulated fixed-point mass first EXCEEDS floor(p * sum)."""
    x = np.asarray(logits, np.float32)
    N = x.shape[0]
    k = N if (top_k <= 0 or top_k > N) else top_k
    order = np.lexsort((np.arange(N), -x.astype(np.float64)))[:k]
    e = _expf32(x[order], x[order[0]], temperature)
    E = _fixed_mass(e)
    kk = k
    if top_p > 1e-7:
        total = int(E.sum(dtype=np.uint64))
        target = int(np.float64(np.float32(top_p)) * np.float64(total))       # (u64)((double)p * (double)sum)
        cum = np.cumsum(E.astype(object))                                      # exact integers
        over = [r for r in range(k) if int(cum[r]) > target]
        if over:
            kk = over[0] + 1
    return order[:kk], e[:kk]


def sample_wide(logits, top_k, top_p, temperature, seed, position=0):
    """-> (token id, margin) of the wide pipeline: score_i = e_i / q_i with u_i keyed by the TOKEN INDEX (0x40000000 + i); the first
    maximum in index order wins."""
    idx, e = wide_final_set(logits, top_k, top_p, temperature)
    u = np.array([uniform01(seed, position, 0x40000000 + int(i)) for i in idx], np.float32)
    with np.errstate(divide="ignore"):
        q = -np.log1p(-u.astype(np.float64))
        score = e.astype(np.float64) / q
    best_score = score.max()
    winners = [int(i) for i, s_ in zip(idx, score) if s_ == best_score]
    rest = score[score < best_score]
    margin = float(best_score / rest.max()) if len(rest) and rest.max() > 0 else np.inf
    return min(winners), margin
