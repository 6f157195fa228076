"""Logits-processor and log-probability oracle in numpy (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Restates, for T = float:
  * cuda::LogitsProcessor, csrc/core/kernel/cuda/beam_search.cu:456-539, kernel by kernel in the reference's order:
      rep_logits_processor          :330-357   (reads a COPY of the scores: :486-492)
      token_count_processor         :373-392   + penalty_logits_processor :359-371
      n_gram_logits_processor       :394-420
      min_length_logits_processor   :423-433
    with the per-request lists of GenerateOp::build_batch_gencfg (generate_op.cpp:239-312).  Every product is one float32 rounding and
    every sum another (no fused multiply-add): `count * frequency` then `+ presence` then `score - total`.
  * logprobs_gpu, csrc/core/operator/generate_opt/generate/generate_impl_gpu.hpp:33-80: log-softmax of the row, the chosen token's
    value (SelectBatchTokenLogprob, kernel/cuda/logprob.cu:15-35), the top_logprobs largest values with their indices.

PARITY: the processors are PINNED on the reference's own device code -- cuda::LogitsProcessor<float> and its kernels sliced from beam_search.cu where
it lies and compiled for gfx950 (oracle/logits_ref.hip, oracle/Makefile `reflogits`): tests/test_gpu_logits_ref.py holds this restatement and the
product equal to them bit for bit.  (Their x86 counterpart cpu::LogitsProcessor, kernel/cpu/beam_search.cpp:343-392, takes ONE GenerateConfig for the
batch and has no frequency penalty: not the path the serving engine's GPU build runs.)  The log-probability half is restated from the source and
checked as mathematics (float64 log-softmax, 2e-5): "parity unpinned" for that half.  The loops below are the kernels' bodies with `tid` iterated
on the host.
"""
import numpy as np

F = np.float32


def logits_processor(score, ids, cur_len, input_len, repetition, frequency, presence, ngram, min_length, eos, suppress):
    """score: f32 [M, N] (a processed copy is returned); ids: int64 [M, max_len]; the rest: per-request sequences of length M."""
    score = np.array(score, dtype=F, copy=True)
    ids = np.asarray(ids, dtype=np.int64)
    M, N = score.shape
    max_len = ids.shape[1]
    for b in range(M):
        cl = int(cur_len[b])
        # ---- repetition penalty (score_in is the copy made at :486-488)
        score_in = score[b].copy()
        p = F(repetition[b])
        for i in range(max_len):
            if suppress[b] != 0 and i < input_len[b]:
                continue
            if i >= cl:
                continue
            t = int(ids[b, i])
            if t < 0 or t >= N:
                continue
            score[b, t] = score_in[t] * p if score_in[t] < 0 else score_in[t] / p
        # ---- frequency / presence penalty over the generated tokens
        count = np.zeros(N, np.int32)
        for i in range(max_len):
            if i < input_len[b] or i >= cl:
                continue
            t = int(ids[b, i])
            if t < 0 or t >= N:
                continue
            count[t] += 1
        total = count.astype(F) * F(frequency[b])
        total = np.where(count > 0, total + F(presence[b]), total).astype(F)
        score[b] = (score[b] - total).astype(F)
        # ---- no-repeat n-gram
        ng = int(ngram[b])
        for i in range(max_len):
            if i < cl and ng > 0 and i + ng - 2 < cl - 1:
                if all(ids[b, i + j] == ids[b, cl - ng + j + 1] for j in range(ng - 1)):
                    t = int(ids[b, i + ng - 1])
                    if 0 <= t < N:   # (the kernel does not check here: an id outside the vocabulary is undefined behaviour in the reference)
                        score[b, t] = F(-1e9)
        # ---- minimum length
        if cl < min_length[b] and 0 <= int(eos[b]) < N:
            score[b, int(eos[b])] = F(-1e9)
    return score


def logprobs(score, chosen, top_n):
    """-> (token_logprob [M] f64, top values [M, top_n] f64, top indices [M, top_n]): value descending, lower index first on ties."""
    x = np.asarray(score, dtype=np.float64)
    m = x.max(axis=1, keepdims=True)
    lp = (x - m) - np.log(np.exp(x - m).sum(axis=1, keepdims=True))
    M, N = x.shape
    tok = np.array([lp[b, int(chosen[b])] for b in range(M)]) if chosen is not None else None
    idx = np.stack([np.lexsort((np.arange(N), -np.asarray(score[b], np.float64)))[:top_n] for b in range(M)]) if top_n else np.zeros((M, 0), np.int64)
    val = np.take_along_axis(lp, idx, axis=1) if top_n else np.zeros((M, 0))
    return tok, val, idx
