"""dihip_logits_processor / dihip_logprobs (csrc/logits_proc.hip: GenerateOp's logits processors and log-probability outputs) against
oracle/logits_proc.py, the restatement of cuda::LogitsProcessor<float> (csrc/core/kernel/cuda/beam_search.cu:456-539) and logprobs_gpu
(generate_impl_gpu.hpp:33-80): processed logits BIT-identical (float32, every product and sum rounded once), log-probabilities within
2e-5 absolute (the kernel sums exp in float64 over a different order than numpy), top-n indices exact including ties."""
import numpy as np
import pytest
import torch

from oracle import logits_proc as lp

pytestmark = pytest.mark.gpu


def _case(rng, M, N, max_len):
    ids = rng.integers(0, N, (M, max_len)).astype(np.int64)
    hot = rng.integers(0, N, 12)
    for b in range(M):                                    # repeats: a small hot set dominates the history
        mask = rng.random(max_len) < 0.6
        ids[b, mask] = hot[rng.integers(0, len(hot), int(mask.sum()))]
    ids[0, 3] = -1                                         # out-of-range ids are skipped (beam_search.cu:349,385)
    ids[min(1, M - 1), 5] = N + 7
    cur = rng.integers(1, max_len + 1, M).astype(np.int32)
    inp = np.minimum(cur, rng.integers(0, max_len // 2 + 1, M)).astype(np.int32)
    cur[0], inp[0] = max_len, 0
    return ids, cur, inp


@pytest.mark.parametrize("M,N,max_len", [(1, 152064, 2048), (7, 32000, 300), (32, 4096, 64), (3, 50, 40)])
def test_processed_logits_are_bit_identical_to_the_oracle(pkg, M, N, max_len):
    from dash_infer_amd import ops
    rng = np.random.default_rng(M * 1000 + max_len)
    ids, cur, inp = _case(rng, M, N, max_len)
    logits = rng.normal(0, 4, (M, N)).astype(np.float32)
    rep = rng.choice([1.0, 1.1, 1.3, 0.8], M).astype(np.float32)
    freq = rng.choice([0.0, 0.1, 0.37, -0.2], M).astype(np.float32)
    pres = rng.choice([0.0, 0.5, 1.2], M).astype(np.float32)
    ng = rng.choice([0, 0, 2, 3, 1], M).astype(np.int32)
    minl = rng.integers(0, max_len + 10, M).astype(np.int32)
    eos = rng.integers(0, N, M).astype(np.int32)
    sup = rng.integers(0, 2, M).astype(np.int32)
    want = lp.logits_processor(logits, ids, cur, inp, rep, freq, pres, ng, minl, eos, sup)
    x = torch.from_numpy(logits).cuda()
    ws = torch.full((M * N * 4,), 0xAB, dtype=torch.uint8, device="cuda")      # the scratch needs no initial state
    for _ in range(2):                                                         # ... and leaves none behind that matters
        x.copy_(torch.from_numpy(logits))
        ops.logits_processor_(x, torch.from_numpy(ids).cuda(), cur, inp, rep, freq, pres, ng, minl, eos, sup, ws=ws)
        torch.cuda.synchronize()
        got = x.cpu().numpy()
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), f"{int((got != want).sum())} logits differ"
    assert (want != logits).any()


def test_neutral_parameters_leave_the_logits_untouched(pkg):
    from dash_infer_amd import ops
    rng = np.random.default_rng(5)
    logits = rng.normal(0, 4, (4, 1000)).astype(np.float32)
    ids = rng.integers(0, 1000, (4, 128)).astype(np.int64)
    x = torch.from_numpy(logits).cuda()
    ops.logits_processor_(x, torch.from_numpy(ids).cuda(), [128, 5, 0, 77], [10, 5, 0, 0])
    torch.cuda.synchronize()
    assert (x.cpu().numpy().view(np.uint32) == logits.view(np.uint32)).all()


def test_workspace_is_checked(pkg):
    from dash_infer_amd import capi, ops
    x = torch.zeros(2, 100, device="cuda")
    with pytest.raises(capi.DihipError):
        ops.logits_processor_(x, torch.zeros(2, 4, dtype=torch.int64, device="cuda"), [1, 1], [0, 0], ws=torch.empty(16, dtype=torch.uint8, device="cuda"))


@pytest.mark.parametrize("M,N,top_n", [(1, 152064, 10), (5, 32000, 5), (3, 700, 32), (2, 6, 10), (4, 1000, 0)])
def test_logprobs_match_the_oracle(pkg, M, N, top_n):
    from dash_infer_amd import ops
    rng = np.random.default_rng(N + top_n)
    logits = rng.normal(0, 3, (M, N)).astype(np.float32)
    logits[0, 1] = logits[0, 4] = logits[0].max() + 0.5          # a tie at the top: the lower index first
    if N > 500:
        logits[M - 1, 100:400] = -1e9                              # masked tokens (the processors' -1e9)
    chosen = rng.integers(0, N, M).astype(np.int64)
    tok, val, idx = ops.logprobs(torch.from_numpy(logits).cuda(), torch.from_numpy(chosen).cuda(), top_n)
    torch.cuda.synchronize()
    wtok, wval, widx = lp.logprobs(logits, chosen, min(top_n, N))
    np.testing.assert_allclose(tok.cpu().numpy(), wtok, atol=2e-5, rtol=0)
    k = min(top_n, N)
    assert (idx.cpu().numpy()[:, :k] == widx).all()
    np.testing.assert_allclose(val.cpu().numpy()[:, :k], wval, atol=2e-5, rtol=0)
    if top_n > N:                                                  # places beyond the row's length say so
        assert (idx.cpu().numpy()[:, N:] == -1).all() and np.isneginf(val.cpu().numpy()[:, N:]).all()
