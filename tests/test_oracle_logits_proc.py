"""oracle/logits_proc.py on hand-worked cases of cuda::LogitsProcessor (csrc/core/kernel/cuda/beam_search.cu:330-539) and of the
log-probability outputs (generate_impl_gpu.hpp:33-80): what each processor touches, in which order, with which arithmetic."""
import numpy as np

from oracle import logits_proc as lp

F = np.float32


def run(score, ids, cur, inp, rep=1.0, freq=0.0, pres=0.0, ng=0, minl=0, eos=0, sup=0):
    return lp.logits_processor(np.asarray([score], F), np.asarray([ids], np.int64), [cur], [inp], [rep], [freq], [pres], [ng], [minl], [eos], [sup])[0]


def test_repetition_penalty_once_per_distinct_token_sign_dependent():
    s = [2.0, -2.0, 1.0, 4.0, 0.5]
    out = run(s, [0, 1, 0, 0, 3, 9, 9], cur=4, inp=2, rep=1.25)       # ids beyond cur_len (3, 9, 9) are not history
    np.testing.assert_array_equal(out, np.array([F(2.0) / F(1.25), F(-2.0) * F(1.25), 1.0, 4.0, 0.5], F))
    out = run(s, [0, 1, 0, 0, 3, 9, 9], cur=5, inp=2, rep=1.25, sup=1)  # suppress: the prompt (ids[:2]) is exempt
    np.testing.assert_array_equal(out, np.array([F(2.0) / F(1.25), -2.0, 1.0, F(4.0) / F(1.25), 0.5], F))


def test_frequency_and_presence_count_generated_tokens_only():
    s = [1.0, 1.0, 1.0, 1.0]
    out = run(s, [0, 0, 1, 1, 1, 2], cur=6, inp=2, freq=0.1, pres=0.5)
    want = np.array([1.0, F(1.0) - (F(3) * F(0.1) + F(0.5)), F(1.0) - (F(1) * F(0.1) + F(0.5)), 1.0], F)   # token 0 occurs in the prompt only
    np.testing.assert_array_equal(out, want)


def test_repetition_then_penalties_compose_in_the_reference_order():
    out = run([3.0, -1.0], [1, 0, 0], cur=3, inp=1, rep=2.0, freq=0.25, pres=1.0)
    np.testing.assert_array_equal(out, np.array([F(3.0) / F(2.0) - (F(2) * F(0.25) + F(1.0)), F(-1.0) * F(2.0)], F))


def test_no_repeat_ngram_bans_the_continuations_of_the_current_suffix():
    # history 5 6 7 5 6 with n = 3: the last two tokens (5 6) were followed by 7 before -> 7 is banned
    out = run(np.zeros(8), [5, 6, 7, 5, 6, 0, 0], cur=5, inp=0, ng=3)
    want = np.zeros(8, F)
    want[7] = -1e9
    np.testing.assert_array_equal(out, want)
    # n = 1: every token of the history is banned (the kernel compares nothing)
    out = run(np.zeros(8), [5, 6, 7, 5, 6, 0, 0], cur=5, inp=0, ng=1)
    assert sorted(np.nonzero(out)[0]) == [5, 6, 7]


def test_min_length_masks_eos_and_runs_last():
    out = run([1.0, 2.0, 3.0], [2, 2], cur=2, inp=0, freq=1.0, minl=5, eos=2)
    np.testing.assert_array_equal(out, np.array([1.0, 2.0, -1e9], F))
    out = run([1.0, 2.0, 3.0], [2, 2], cur=2, inp=0, minl=2, eos=2)
    np.testing.assert_array_equal(out, np.array([1.0, 2.0, 3.0], F))


def test_neutral_parameters_change_nothing_and_out_of_range_ids_are_skipped():
    rng = np.random.default_rng(0)
    s = rng.normal(0, 3, 50).astype(F)
    np.testing.assert_array_equal(run(s, [3, 70, -1, 4, 4], cur=5, inp=1), s)
    out = run(s, [3, 70, -1, 4, 4], cur=5, inp=1, rep=1.5, pres=0.25)
    changed = np.nonzero(out != s)[0]
    assert sorted(changed) == [3, 4]


def test_logprobs_values_and_tie_order():
    x = np.log(np.array([[0.1, 0.4, 0.4, 0.1]]))
    tok, val, idx = lp.logprobs(x.astype(F), [2], 3)
    np.testing.assert_allclose(tok, [np.log(0.4)], rtol=1e-6)
    assert list(idx[0]) == [1, 2, 0]
    np.testing.assert_allclose(val[0], np.log([0.4, 0.4, 0.1]), rtol=1e-6)
