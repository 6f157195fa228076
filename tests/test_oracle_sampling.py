"""oracle/sampling.py against the reference's x86 sampling pipeline semantics (generate_impl_cpu.hpp:120-170, topp.cpp, sample.cpp)
on cases worked out by hand, and the statistical contract: the exponential race draws token i with probability prob_i."""
import numpy as np

from oracle import sampling


def test_topk_then_topp_prefix_rule():
    logits = np.log(np.array([0.5, 0.2, 0.15, 0.1, 0.05], np.float64)).astype(np.float32)
    idx, p = sampling.final_probs(logits, top_k=4, top_p=0.6, temperature=1.0)
    # top-4 probabilities renormalised: .5/.95, .2/.95, ... ; cumulated .526, .737 -> the first rank whose sum EXCEEDS 0.6 is rank 1
    assert list(idx) == [0, 1]
    np.testing.assert_allclose(p, [0.5 / 0.7, 0.2 / 0.7], rtol=1e-6)
    idx, p = sampling.final_probs(logits, top_k=0, top_p=0.0, temperature=1.0)      # p <= 1e-7: no cut; k = 0: everything (<= 1024)
    assert list(idx) == [0, 1, 2, 3, 4]
    np.testing.assert_allclose(p.sum(), 1.0, rtol=1e-12)
    idx, p = sampling.final_probs(logits, top_k=3, top_p=1.0, temperature=0.5)      # nothing exceeds 1: all k stay; T sharpens
    w = np.array([0.5, 0.2, 0.15]) ** 2
    np.testing.assert_allclose(p, w / w.sum(), rtol=1e-5)


def test_ties_take_the_lowest_index_and_greedy_is_argmax():
    logits = np.array([1.0, 3.0, 3.0, 2.0, 3.0], np.float32)
    idx, _ = sampling.candidates(logits, 2)
    assert list(idx) == [1, 2]
    tok, _ = sampling.sample(logits, top_k=1, top_p=1.0, temperature=1.0, seed=5, position=9)
    assert tok == 1


def test_uniform_stream_is_a_pure_function_in_the_unit_interval():
    u = [float(sampling.uniform01(42, 7, r)) for r in range(2000)]
    assert all(0.0 <= x < 1.0 for x in u) and len(set(u)) > 1990
    assert abs(np.mean(u) - 0.5) < 0.03
    assert sampling.uniform01(42, 7, 3) == sampling.uniform01(42, 7, 3) != sampling.uniform01(42, 8, 3)


def test_the_exponential_race_draws_with_the_final_probabilities():
    probs = np.array([0.55, 0.25, 0.15, 0.05])
    logits = np.log(probs).astype(np.float32)
    counts = np.zeros(4)
    n = 6000
    for pos in range(n):
        tok, _ = sampling.sample(logits, top_k=4, top_p=1.0, temperature=1.0, seed=1234, position=pos)
        counts[tok] += 1
    np.testing.assert_allclose(counts / n, probs, atol=0.02)


def test_wide_rows_whole_vocabulary_top_p_and_fixed_point_prefix():
    """top_k == 0 = the whole vocabulary (generate_op.cpp:338-339); the wide restatement picks the same final set as the sorted pipeline
    (fixed-point masses only move a cut that sits within 2^-32 of the target) and draws with the final probabilities."""
    rng = np.random.default_rng(3)
    x = rng.normal(0, 2, 3000).astype(np.float32)
    for k, p, T in ((0, 0.9, 1.0), (0, 1.0, 1.0), (2000, 0.5, 0.7), (0, 0.0, 1.3)):
        idx_w, e = sampling.wide_final_set(x, k, p, T)
        idx_n, p2 = sampling.final_probs(x, k if k else len(x), p, T)
        assert list(idx_w) == list(idx_n)
        np.testing.assert_allclose(e / e.sum(), p2, rtol=1e-5)
    assert len(sampling.candidates(x, 0)[0]) == len(x)               # k = 0: every token is a candidate
    small = np.log(np.array([0.5, 0.3, 0.2], np.float32))
    draws = [sampling.sample_wide(small, 0, 1.0, 1.0, 11, pos)[0] for pos in range(3000)]
    np.testing.assert_allclose(np.bincount(draws, minlength=3) / 3000, [0.5, 0.3, 0.2], atol=0.04)
