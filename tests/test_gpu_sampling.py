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
    cfgs = [(1, 1.0, 1.0), (50, 0.9, 0.8), (0, 0.95, 1.3), (1024, 1.0, 1.0), (7, 0.0, 0.5), (200, 0.3, 2.0)]
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


@pytest.mark.parametrize("fuse", [True, False])
def test_sampling_requests_through_the_operator_list(pkg, fuse):
    """Two requests, one greedy and one sampling (top_k 40, top_p 0.9, T 0.8), through the model runner: the sampled ids are what
    the oracle draws from the operator's own logits at the request's position; under the fused list the captured step (positions
    read on the device) draws exactly what eager stepping (positions staged from the host) draws."""
    from dash_infer_amd import decoder
    from tests.test_gpu_host_runner import Host, SMALL
    cfg = decoder.ModelConfig("sampling-test", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=5, keep_fp=True)
    rng = np.random.default_rng(1)
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in (12, 7)]
    gen = [dict(), dict(top_k=40, top_p=0.9, temperature=0.8, seed=4242)]

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
    tok, margin = sampling.sample(lo0[1], 40, 0.9, 0.8, 4242, len(prompts[1]))
    if margin > 1 + 1e-4:
        assert ids[0][1] == tok
        decided += 1
    for t in range(5):
        assert ids[t + 1][0] == int(np.argmax(steps[t][0]))
        tok, margin = sampling.sample(steps[t][1], 40, 0.9, 0.8, 4242, len(prompts[1]) + t + 1)
        if margin > 1 + 1e-4:
            assert ids[t + 1][1] == tok, f"step {t}"
            decided += 1
    assert decided >= 4
    if fuse:
        ids_eager, _, _ = run(graph=False)
        assert ids_eager == ids
