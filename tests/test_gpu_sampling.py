"""dihip_sample (csrc/sample.hip: the sampling half of GenerateOp) against oracle/sampling.py -- the reference x86 pipeline
(generate_impl_cpu.hpp:120-170: top-k -> softmax(T) -> top-p prefix -> softmax -> exponential race) with the backend's counter-based
random stream restated bit for bit: candidate sets and their order exact, final probabilities to float accuracy, the drawn token
equal wherever the oracle's best / second-best score ratio is not within rounding of 1; greedy rows, ties, tiny vocabularies, the
device-resident position counters; and through the operator layer (GenerateOp on HIP, fused DihipGreedy under graph replay)."""
import numpy as np
import pytest
import torch

from oracle import sampling

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [152064, 4096, 700, 10])
def test_candidates_probabilities_and_draws_match_the_oracle(pkg, N):
    from dash_infer_amd import ops
    rng = np.random.default_rng(N)
    cfgs = [(1, 1.0, 1.0), (50, 0.9, 0.8), (1000, 0.95, 1.3), (1024, 1.0, 1.0), (7, 0.0, 0.5), (200, 0.3, 2.0)]
    M = len(cfgs)
    logits = (rng.normal(0, 2.5, (M, N))).astype(np.float32)
    logits[1, 5] = logits[1, 77 % N] = logits[1].max() + 1.0       # a tie at the top
    if N > 2000:
        logits[3, 1000:1900] = 0.25                                 # a long run of ties straddling the k-th place
    seeds = [int(s) for s in rng.integers(0, 2 ** 62, M)]
    pos = torch.tensor([3, 2049, 0, 77, 123456, 9], dtype=torch.int32, device="cuda")
    ids, probs, cand = ops.sample(torch.from_numpy(logits).cuda(), [c[0] for c in cfgs], [c[1] for c in cfgs], [c[2] for c in cfgs], seeds,
                                  position=pos, want_probs=True)
    torch.cuda.synchronize()
    ids, probs, cand = ids.cpu().numpy(), probs.cpu().numpy(), cand.cpu().numpy()
    decided = 0
    for m, (k, p, T) in enumerate(cfgs):
        want_idx, want_v = sampling.candidates(logits[m], k)
        kk = len(want_idx)
        assert list(cand[m, :kk]) == list(want_idx), f"row {m}: candidate order"
        assert (cand[m, kk:] == -1).all()
        fidx, fp = sampling.final_probs(logits[m], k, p, T)
        np.testing.assert_allclose(probs[m, :len(fidx)], fp, rtol=2e-5, atol=1e-9, err_msg=f"row {m}")
        assert (probs[m, len(fidx):] == 0).all(), f"row {m}: the top-p prefix is {len(fidx)} long"
        tok, margin = sampling.sample(logits[m], k, p, T, seeds[m], int(pos[m]))
        assert int(ids[m]) in set(int(i) for i in fidx)
        if margin > 1.0 + 1e-4:
            assert int(ids[m]) == tok, f"row {m}: drew {ids[m]}, oracle {tok} (margin {margin:.6f})"
            decided += 1
    assert decided >= 4
    assert int(ids[0]) == int(np.argmax(logits[0]))                 # top_k = 1: greedy, lowest index on ties


def test_draw_frequencies_follow_the_final_probabilities_and_counters_advance(pkg):
    from dash_infer_amd import ops
    probs = np.array([0.5, 0.3, 0.15, 0.05])
    M = 2048
    logits = torch.from_numpy(np.tile(np.log(probs).astype(np.float32), (M, 1))).cuda()
    a = torch.arange(M, dtype=torch.int32, device="cuda")
    b = torch.zeros(M, dtype=torch.int32, device="cuda")
    ids = ops.sample(logits, [4] * M, [1.0] * M, [1.0] * M, [99] * M, position=a, advance=(a, b))
    torch.cuda.synchronize()
    freq = np.bincount(ids.cpu().numpy(), minlength=4) / M
    np.testing.assert_allclose(freq, probs, atol=0.035)
    assert torch.equal(a.cpu(), torch.arange(1, M + 1, dtype=torch.int32)) and int(b.sum()) == M
    # the stream is a pure function of (seed, position, rank): same inputs, same draws; another seed, other draws
    a2 = torch.arange(M, dtype=torch.int32, device="cuda")
    again = ops.sample(logits, [4] * M, [1.0] * M, [1.0] * M, [99] * M, position=a2)
    other = ops.sample(logits, [4] * M, [1.0] * M, [1.0] * M, [100] * M, position=a2)
    assert torch.equal(ids, again) and not torch.equal(ids, other)


@pytest.mark.parametrize("fuse,top_k", [(True, 40), (False, 40), (True, 0), (False, 0)])
def test_sampling_requests_through_the_operator_list(pkg, fuse, top_k):
    """Two requests, one greedy and one sampling (top_k 40 -- or 0, the whole vocabulary: the wide kernel -- top_p 0.9, T 0.8), through the
    model runner: the sampled ids are what the oracle draws from the operator's own logits at the request's position; under the fused list
    the captured step (positions read on the device) draws exactly what eager stepping (positions staged from the host) draws."""
    from dash_infer_amd import decoder
    from tests.test_gpu_host_runner import Host, SMALL
    cfg = decoder.ModelConfig("sampling-test", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=5, keep_fp=True)
    rng = np.random.default_rng(1)
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in (12, 7)]
    gen = [dict(), dict(top_k=top_k, top_p=0.9, temperature=0.8, seed=4242)]
    draw = sampling.sample if 1 <= top_k <= 1024 else sampling.sample_wide

    def run(graph):
        h = Host(model, 2, 64, 16, "none", fuse=fuse)
        out, lo = [], []
        firsts = []
        for pr, g in zip(prompts, gen):
            k, v = h.spans()
            firsts.append(h.start(pr, k, v, **g))
            lo.append(h.logits().float().cpu().numpy()[0])
        out.append(firsts)
        step_logits = []
        for _ in range(5):
            out.append(h.steps(1, graph=graph))
            step_logits.append(h.logits().float().cpu().numpy())
        h.close()
        return out, lo, step_logits

    ids, lo0, steps = run(graph=fuse)
    # the greedy request is the arg-max of its logits; the sampling request's draw is the oracle's at position = sequence length
    assert ids[0][0] == int(np.argmax(lo0[0]))
    decided = 0
    tok, margin = draw(lo0[1], top_k, 0.9, 0.8, 4242, len(prompts[1]))
    if margin > 1 + 1e-3:
        assert ids[0][1] == tok
        decided += 1
    for t in range(5):
        assert ids[t + 1][0] == int(np.argmax(steps[t][0]))
        tok, margin = draw(steps[t][1], top_k, 0.9, 0.8, 4242, len(prompts[1]) + t + 1)
        if margin > 1 + 1e-3:
            assert ids[t + 1][1] == tok, f"step {t}"
            decided += 1
    assert decided >= 4
    if fuse:
        ids_eager, _, _ = run(graph=False)
        assert ids_eager == ids


def _order_key(v):
    b = np.float32(v).view(np.uint32)
    return int(~b & 0xFFFFFFFF) if b & 0x80000000 else int(b | 0x80000000)


@pytest.mark.parametrize("N", [152064, 5000, 1500])
def test_top_k_zero_and_beyond_1024_run_the_wide_kernel(pkg, N):
    """top_k == 0 is "the whole vocabulary" in the reference (generate_op.cpp:338-339: pure top-p sampling) and its default build
    serves any k.  Rows with k == 0 or k > 1024 run sample_wide_kernel: the same pipeline without the sorted candidate list -- fixed-point
    masses, the top-p prefix by a radix select on mass, the race keyed by token index -- against oracle/sampling.py's restatement: the final
    candidate set (threshold key + ties at it) where the cut is not within rounding of a boundary, the drawn token where the race is decided
    by more than rounding; narrow rows in the same call are unaffected."""
    from dash_infer_amd import ops
    rng = np.random.default_rng(N + 1)
    cfgs = [(0, 0.9, 1.0), (0, 1.0, 0.7), (0, 0.0, 1.0), (1025, 0.95, 1.0), (N, 0.5, 1.5), (4000, 0.0, 1.0), (40, 0.9, 0.8), (0, 0.3, 2.0), (2000, 0.999, 1.0)]
    M = len(cfgs)
    logits = rng.normal(0, 2.0, (M, N)).astype(np.float32)
    logits[0, [3, 30, 300]] = logits[0].max() + 0.5                 # ties at the top of a pure top-p row
    logits[3, 100:1400] = 1.75                                       # a run of ties straddling the 1025-th place
    logits[7, ::2] = logits[7, 0]                                    # half the vocabulary tied
    seeds = [int(s_) for s_ in rng.integers(0, 2 ** 62, M)]
    pos = torch.tensor(list(range(5, 5 + M)), dtype=torch.int32, device="cuda")
    ids, probs, cand = ops.sample(torch.from_numpy(logits).cuda(), [c[0] for c in cfgs], [c[1] for c in cfgs], [c[2] for c in cfgs], seeds,
                                  position=pos, want_probs=True)
    torch.cuda.synchronize()
    ids, cand = ids.cpu().numpy(), cand.cpu().numpy()
    decided = exact_sets = 0
    for m, (k, p, T) in enumerate(cfgs):
        if 1 <= k <= 1024:                                           # a narrow row beside the wide ones
            tok, margin = sampling.sample(logits[m], k, p, T, seeds[m], int(pos[m]))
            assert margin <= 1.0 + 1e-4 or int(ids[m]) == tok
            continue
        idx, e = sampling.wide_final_set(logits[m], k, p, T)
        # the kernel's final set as (threshold key, ties at it) -> its size
        thr, ties = int(np.uint32(cand[m, 0])), int(np.uint32(cand[m, 1]))
        keys = np.array([_order_key(v) for v in logits[m]], dtype=np.uint64)
        got_size = int((keys > thr).sum()) + min(ties, int((keys == thr).sum()))
        # how far the cumulated mass at the cut is from the target, relative to the total: a device expf one ulp off moves a boundary only if tiny
        kk_full = N if (k <= 0 or k > N) else k
        order = np.lexsort((np.arange(N), -logits[m].astype(np.float64)))[:kk_full]
        E = sampling._fixed_mass(sampling._expf32(logits[m][order], logits[m][order[0]], T)).astype(object)
        total, cum = int(E.sum()), np.cumsum(E)
        target = int(np.float64(np.float32(p)) * np.float64(total))
        near = p > 1e-7 and len(idx) < kk_full and min(abs(int(cum[len(idx) - 1]) - target), abs(int(cum[max(len(idx) - 2, 0)]) - target)) < 2e-6 * total
        if not near:
            assert got_size == len(idx), f"row {m} (k {k}, p {p}): final set of {got_size}, oracle {len(idx)}"
            exact_sets += 1
        tok, margin = sampling.sample_wide(logits[m], k, p, T, seeds[m], int(pos[m]))
        assert int(ids[m]) in set(int(i) for i in order)
        if margin > 1.0 + 1e-3 and not near:
            assert int(ids[m]) == tok, f"row {m} (k {k}, p {p}): drew {ids[m]}, oracle {tok} (margin {margin:.6f})"
            decided += 1
    assert decided >= 4 and exact_sets >= 5, (decided, exact_sets)


def test_wide_rows_draw_with_the_final_probabilities(pkg):
    """top_k = 0, top_p = 1: the race over the whole (small) vocabulary draws token i with probability softmax(x / T)_i"""
    from dash_infer_amd import ops
    probs = np.array([0.4, 0.25, 0.2, 0.1, 0.05])
    M = 4096
    logits = torch.from_numpy(np.tile(np.log(probs).astype(np.float32), (M, 1))).cuda()
    a = torch.arange(M, dtype=torch.int32, device="cuda")
    ids = ops.sample(logits, [0] * M, [1.0] * M, [1.0] * M, [7] * M, position=a)
    torch.cuda.synchronize()
    freq = np.bincount(ids.cpu().numpy(), minlength=5) / M
    np.testing.assert_allclose(freq, probs, atol=0.03)
    # top_p = 0.6 keeps the shortest prefix EXCEEDING 0.6: {0.4, 0.25} -> renormalised 0.615 / 0.385
    ids = ops.sample(logits, [0] * M, [0.6] * M, [1.0] * M, [8] * M, position=a)
    freq = np.bincount(ids.cpu().numpy(), minlength=5) / M
    np.testing.assert_allclose(freq, [0.4 / 0.65, 0.25 / 0.65, 0, 0, 0], atol=0.03)
