"""GenerateOp's logits processors and log-probability outputs THROUGH THE OPERATOR LAYER (host/sampling_host.h LogitsProcParams over
csrc/logits_proc.hip), both forms, against oracle/logits_proc.py (cuda::LogitsProcessor<float>, beam_search.cu:456-539; logprobs_gpu,
generate_impl_gpu.hpp:33-80):

  * staged form -- GenerateOp on HIP driven operator by operator with the reference's request plumbing (PreProcessId creates the host
    "generated_ids", GenerateOp stages it every Forward like fill_max_dec_ids, UpdateProbs' lists fill): the chosen ids are the arg-max of
    the ORACLE-processed logits, the processed logits (in place, as the reference) are bit-identical, the lists carry the oracle's values;
  * rows form -- the model runner's fused list under hipGraph REPLAY: a request with processors + logprobs decodes with its token history
    and record log device-resident (nothing from the host per step).  A second, neutral host is teacher-forced along the same tokens
    (stop + adopt per step) to provide the raw logits of every step; oracle(raw logits, history) must reproduce the first host's processed
    logits bit for bit, its ids, and its log-probability records."""
import numpy as np
import pytest
import torch

from oracle import logits_proc as lp

pytestmark = pytest.mark.gpu

GEN = dict(repetition_penalty=1.3, frequency_penalty=0.25, presence_penalty=0.5, no_repeat_ngram_size=2, min_length=9, eos_token_id=3,
           logprobs=True, top_logprobs=4)


def oracle_step(raw, history, input_len, g):
    ids = np.asarray([history], np.int64)
    return lp.logits_processor(raw[None, :], ids, [len(history)], [input_len], [g.get("repetition_penalty", 1.0)], [g.get("frequency_penalty", 0.0)],
                               [g.get("presence_penalty", 0.0)], [g.get("no_repeat_ngram_size", 0)], [g.get("min_length", 0)],
                               [g.get("eos_token_id", -1)], [int(g.get("suppress_repetition_in_generation", False))])[0]


def test_generate_op_staged_form_operator_by_operator(pkg):
    from dash_infer_amd import hostapi, ops
    V, L, max_len = 3000, 6, 64
    rng = np.random.default_rng(3)
    m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16, max_batch=2, max_len=max_len)
    pre = m.create_op("PreProcessId", "preprocess_id", ["input_ids"], ["pre.out"])
    gen = m.create_op("GenerateOp", "generate", ["logits"], ["generated_ids_out"], [], "top_k=i:1")
    prompt = [int(t) for t in rng.integers(0, V, L)]
    prompt[2] = prompt[0]
    g = dict(GEN, min_length=L + 3)
    m.set_runtime(True, [0], [[[]]], [[[]]])
    m.request_attach(0, prompt, max_length=max_len, eos=3)
    m.request_generation(0, L, **g)
    m.forward(pre)
    history = list(prompt)
    logits_t = torch.empty(1, 1, V, dtype=torch.float32, device="cuda")
    m.set_tensor("logits", logits_t, "f32")
    for t in range(6):
        raw = rng.normal(0, 3, V).astype(np.float32)
        raw[history[-1]] += 6.0                      # make the penalties matter: the last token would win again
        logits_t.copy_(torch.from_numpy(raw).view(1, 1, V))
        if t == 0:
            m.request_set_step(0, 0, in_length_bias=L)     # context phase: cur_len = step + in_length_bias = L
        else:
            m.set_phase(False)
            m.request_set_step(0, L + t, in_length_bias=0)
        m.reshape(gen)
        m.forward(gen)
        torch.cuda.synchronize()
        want = oracle_step(raw, history, L, g)
        got = logits_t.view(-1).cpu().numpy()
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), f"step {t}: processed logits"
        _, shp, ptr = m.get_tensor("generated_ids_out")
        from tests.test_gpu_host_graph import view_of
        tok = int(view_of(ptr, [1], torch.int64).cpu()[0])
        assert tok == int(np.argmax(want)), f"step {t}"
        assert tok != 3 or len(history) >= g["min_length"]
        m.request_put_token(0, len(history), tok)          # fill_generated_ids
        history.append(tok)
        wtok, wval, widx = lp.logprobs(want[None, :], [tok], 4)
        gtok, gval, gidx = m.request_logprobs(0, t, 1, 4, runner=False)
        assert gidx == [list(map(int, widx[0]))]
        np.testing.assert_allclose(gtok, wtok, atol=2e-5)
        np.testing.assert_allclose(gval[0], wval[0], atol=2e-5)
    m.close()


def test_processors_and_logprobs_replay_in_the_captured_step(pkg):
    from dash_infer_amd import decoder
    from tests.test_gpu_host_runner import Host, SMALL
    cfg = decoder.ModelConfig("extras-test", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=11, keep_fp=True)
    rng = np.random.default_rng(2)
    prompt = [int(t) for t in rng.integers(0, cfg.vocab, 9)]
    prompt[4] = prompt[1]
    L, n_steps = len(prompt), 8
    g = dict(GEN, min_length=L + 4)

    h1 = Host(model, 1, 64, 16, "none", fuse=True)
    assert h1.report["fused"]
    k1, v1 = h1.spans()
    with torch.cuda.stream(h1.stream):
        h1.m.next_request_generation(**g)
    ids1 = [h1.start(prompt, k1, v1)]
    proc = [h1.logits().float().cpu().numpy()[0].copy()]
    for _ in range(n_steps):
        ids1.append(h1.steps(1, graph=True)[0])
        proc.append(h1.logits().float().cpu().numpy()[0].copy())
    with torch.cuda.stream(h1.stream):
        rtok, rval, ridx = h1.m.request_logprobs(0, L, n_steps + 1, 4, runner=True)
    h1.close()

    # the raw logits of the same token sequence: a neutral host, teacher-forced (stop + adopt with H1's token before every step)
    h0 = Host(model, 1, 64, 16, "none", fuse=True)
    k0, v0 = h0.spans()
    h0.start(prompt, k0, v0)
    raw = [h0.logits().float().cpu().numpy()[0].copy()]
    for t in range(n_steps):
        with torch.cuda.stream(h0.stream):
            h0.m.request_stop(0)
            h0.m.request_adopt(L + t, ids1[t], k0, v0)
        h0.steps(1, graph=False)
        raw.append(h0.logits().float().cpu().numpy()[0].copy())
    h0.close()

    history = list(prompt)
    changed = 0
    for t in range(n_steps + 1):
        if t > 0:
            history.append(ids1[t - 1])                     # the step's input id joins the history on the device
        want = oracle_step(raw[t], history, L, g)
        assert (proc[t].view(np.uint32) == want.view(np.uint32)).all(), f"step {t}: processed logits differ from oracle(raw logits, history)"
        assert ids1[t] == int(np.argmax(want)), f"step {t}"
        changed += int(np.argmax(want) != np.argmax(raw[t]))
        wtok, wval, widx = lp.logprobs(want[None, :], [ids1[t]], 4)
        assert ridx[t] == list(map(int, widx[0])), f"step {t}: record {L + t}"
        np.testing.assert_allclose(rtok[t], wtok[0], atol=2e-5)
        np.testing.assert_allclose(rval[t], wval[0], atol=2e-5)
    assert 3 not in ids1[:3]                                # min_length masks eos while cur_len < min_length
    assert changed >= 1, "the processors never changed a choice: the test would not notice their absence"


def test_processors_without_a_history_are_refused_not_ignored(pkg):
    from dash_infer_amd import decoder, hostapi
    from tests.test_gpu_host_runner import Host, SMALL
    cfg = decoder.ModelConfig("extras-test", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=11, keep_fp=True)
    h = Host(model, 1, 64, 16, "none", fuse=True)
    k, v = h.spans()
    with torch.cuda.stream(h.stream):
        h.m.next_request_generation(repetition_penalty=1.2)
        with pytest.raises(hostapi.HostError):              # a cached prefix carries no ids: refused at StartRequest
            h.m.request_start([1, 2, 3], k, v, prefix_len=16)
    h.close()
