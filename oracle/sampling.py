"""Sampling oracle in numpy (TEST INFRASTRUCTURE ONLY): the x86 path of GenerateOp's sampling half,
csrc/core/operator/generate_opt/generate/generate_impl_cpu.hpp:120-170 (gen_sample_cpu):
    cpu::TopKKernel (k largest logits, descending)
    -> cpu::SoftmaxKernel(values, k, temperature)          probabilities of the k candidates
    -> cpu::TopPKernel (kernel/cpu/topp.cpp:14-30)          k <- 1 + first rank whose cumulated probability EXCEEDS p (p > 1e-7)
    -> cpu::SoftmaxKernel over the first k candidates again
    -> cpu::SampleKernel (kernel/cpu/sample.cpp:42-68)      score_i = prob_i / q_i, q_i = -log1p(-u_i); the first maximum wins
PARITY UNPINNED for the random stream: the reference draws u from std::mt19937 (x86) or Philox (CUDA), which no other device
reproduces; the HIP backend defines its own counter-based stream (csrc/sample.hip: 24 high bits of a splitmix64 hash of
(seed, position, rank)), restated here bit for bit.  top_k <= 0 or > 1024 -> 1024 (CONFIG_SAMPLE_CONSTRAIN_MAX_K)."""
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
    k = top_k if 0 < top_k <= 1024 else 1024
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
