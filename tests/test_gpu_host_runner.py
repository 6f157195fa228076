"""The C++ model runner + fusion pass of the HIP operator layer: the reference's Qwen2 operator list
(tests/ref_graph.py) goes through host/fusion_pass.cpp, every fused operator is created by OpFactory for DeviceType::HIP and
driven Reshape -> Alloc -> Forward by host/model_runner.cpp the way AsModel drives its operators (csrc/core/model/model.cpp:
566-650 context phase, :1212-1325 decoder phase), the decoder step captured once as a hipGraph and replayed.

  * the fused list's logits are BIT-IDENTICAL to decoder.DecodeSession (the Python runner bench.py times) -- context phase and
    every decode step, 16-bit / int8 / uint4 caches, the GEMV and the small-batch GEMM families (norm hand-on, FRAG32 layouts);
  * continuous batching: requests join and leave the running batch, every request decodes the tokens it decodes alone;
  * a prefill over a cached prefix (prefix cache hit) through the UNFUSED operators (Rotary + DecOptMQA: ADVICE r3, positions
    step + t) and through the fused ones gives the logits of the same prompt prefilled from scratch."""
import numpy as np
import pytest
import torch

from tests import ref_graph
from tests.test_gpu_host_graph import view_of

pytestmark = pytest.mark.gpu

SMALL = dict(hidden=512, layers=2, n_heads=4, n_kv=2, head_dim=128, inter=1024, vocab=2048)
WIDE = dict(hidden=1024, layers=3, n_heads=8, n_kv=4, head_dim=128, inter=2048, vocab=4096)
KV = {"none": 0, "i8": 1, "u4": 2}


class Host:
    """hostapi.Model + the reference graph + a span pool, on its own stream (a captured step needs a non-NULL stream)."""

    def __init__(self, model, batch, max_len, span, kv_mode, fuse=True, ft="bf16", exported=False, serialized=None):
        from dash_infer_amd import hostapi, ops
        cfg = model.cfg
        self.cfg, self.nl, self.spr = cfg, len(model.layers), (max_len + span - 1) // span
        self.stream = torch.cuda.Stream()
        torch.cuda.synchronize()
        self.pool = ops.SpanPool(2 * batch * self.nl * self.spr + 4, cfg.n_kv, span, cfg.head_dim, kv_mode,
                                 torch.float16 if ft == "f16" else torch.bfloat16)
        with torch.cuda.stream(self.stream):
            self.m = hostapi.Model(ops.cur_stream(), cfg.n_heads, cfg.n_kv, cfg.head_dim, span, KV[kv_mode], max_batch=batch, max_len=max_len)
            ref_graph.register_weights(self.m, model, ft)
            self.graph = ref_graph.qwen2_graph(self.nl, model.quant.wbits, model.quant.group, cfg.eps, cfg.n_heads, cfg.n_kv, cfg.rope_theta,
                                               moe=(cfg.moe.num_experts, cfg.moe.top_k) if cfg.moe is not None else None)
            if exported:   # the arities the reference's converter writes + gen_graph's UpdateId (ref_graph.as_exported)
                self.graph = ref_graph.as_exported(self.graph)
            if serialized is not None:   # bytes of an allspark TransformerProto: through the C++ wire reader (host/graph_wire.h)
                self.m.graph_add_serialized(serialized)
            else:
                ref_graph.add_graph(self.m, self.graph)
            self.report = self.m.graph_build(fuse=fuse)
        self.stream.synchronize()

    def spans(self):
        return ([[self.pool.alloc()[0] for _ in range(self.spr)] for _ in range(self.nl)],
                [[self.pool.alloc()[0] for _ in range(self.spr)] for _ in range(self.nl)])

    def start(self, prompt, k, v, **kw):
        with torch.cuda.stream(self.stream):
            return self.m.request_start(prompt, k, v, **kw)

    def steps(self, n=1, graph=True):
        with torch.cuda.stream(self.stream):
            self.m.decode_steps(n, graph=graph)
            return self.m.sync_ids()

    def logits(self):
        _, shp, ptr = self.m.get_tensor("logits")
        return view_of(ptr, shp, torch.float32 if self.report["fused"] else torch.bfloat16).reshape(-1, shp[-1])

    def close(self):
        self.m.close()


@pytest.mark.parametrize("shape,wbits,group,kv_mode,batch,gptq", [
    (SMALL, 4, 128, "none", 1, False),    # the headline path: GEMV kernels, Rotary + append + attention in one launch
    (WIDE, 4, 128, "u4", 32, True),       # configs[2] family: small-batch GEMMs with the norm handed on, FRAG32, quantising append
    (WIDE, 8, -1, "i8", 3, False),        # int8 per-channel, int8 cache, GEMV kernels at M = 3
    (WIDE, 4, 128, "none", 17, False),    # odd batch on the small-batch kernels, 16-bit cache
    (WIDE, 8, 128, "none", 8, False),     # int8 sub-channel
])
def test_fused_operator_list_is_bit_identical_to_decode_session(pkg, shape, wbits, group, kv_mode, batch, gptq):
    from dash_infer_amd import decoder
    cfg = decoder.ModelConfig("runner-test", **shape)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(wbits, group, gptq_like_zeros=gptq), seed=909, keep_fp=True)
    span, max_len, steps = 16, 96, 6
    rng = np.random.default_rng(batch + wbits)
    lens = [int(x) for x in rng.integers(3, 40, batch)]
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in lens]
    # ---- the Python runner (what bench.py times by default)
    sess = decoder.DecodeSession(model, batch, max_len=max_len, span_len=span, kv_mode=kv_mode)
    lo0 = sess.prefill(prompts).clone()
    ids0 = sess.ids.cpu().tolist()
    want = []
    for _ in range(steps):
        sess.step()
        torch.cuda.synchronize()
        want.append((sess.logits.clone(), sess.ids.cpu().tolist()))
    # ---- the C++ runner over the fused operator list
    h = Host(model, batch, max_len, span, kv_mode)
    assert h.report["fused"] and h.report["layers"] == len(model.layers), h.report["why"]
    assert "DihipRopeSpanAttn" in h.report["types"] and "LayerNormNoBeta" not in h.report["types"]
    for b, pr in enumerate(prompts):
        k, v = h.spans()
        first = h.start(pr, k, v)
        assert first == ids0[b], f"request {b}: first id {first} != {ids0[b]}"
        assert torch.equal(h.logits()[0], lo0[b]), f"request {b}: context-phase logits are not bit-identical"
    for t in range(steps):
        ids = h.steps(1, graph=True)          # step 0 captures, the others replay
        assert ids == want[t][1], f"step {t}: ids differ"
        assert torch.equal(h.logits(), want[t][0]), f"step {t}: logits are not bit-identical to DecodeSession"
    h.close()


MOE_SMALL = dict(hidden=512, layers=2, n_heads=4, n_kv=2, head_dim=128, inter=512, vocab=2048)


@pytest.mark.parametrize("group,batch,experts,top_k", [
    (-1, 1, 8, 2),      # one request: GEMV kernels, routing and expert launches apart
    (128, 3, 8, 2),     # grouped routing + combine (the 8-launch block), row-major shared-expert activations
    (-1, 8, 16, 4),     # + the shared expert's SwiGLU output in FRAG32 for its down projection
])
def test_mixture_of_experts_list_is_bit_identical_to_decode_session(pkg, group, batch, experts, top_k):
    """BASELINE configs[4] architecture through the operator API: the reference's MoE layer list (qwen_v20_moe.py:318-391,
    tests/ref_graph.py) -> fusion pass -> DihipMoeBlock, against decoder.DecodeSession (context phase and decode steps, bit for
    bit) and the unfused list (MOEA16W8, CalcExpert, UnaryGLU, Gemm ... one launch per reference operator) within bf16 rounding;
    the context phase against the oracle decoding the prompt token by token."""
    from dash_infer_amd import decoder
    from tests.test_gpu_decoder import oracle_of
    cfg = decoder.ModelConfig("moe-runner-test", **MOE_SMALL, moe=decoder.MoEConfig(experts, top_k, 256))
    model = decoder.build_random_model(cfg, decoder.QuantSpec(8, group), seed=1357, keep_fp=True)
    span, max_len, steps = 16, 64, 5
    rng = np.random.default_rng(batch + experts)
    lens = [int(x) for x in rng.integers(3, 24, batch)]
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in lens]
    sess = decoder.DecodeSession(model, batch, max_len=max_len, span_len=span, kv_mode="none")
    lo0 = sess.prefill(prompts).clone()
    ids0 = sess.ids.cpu().tolist()
    want = []
    for _ in range(steps):
        sess.step()
        torch.cuda.synchronize()
        want.append((sess.logits.clone(), sess.ids.cpu().tolist()))
    # the context phase of request 0 against the oracle (numpy, independent of the GPU) fed the prompt one token at a time
    ref = oracle_of(model, "none")
    for tok in prompts[0]:
        lo_ref = ref.step(np.asarray([tok]))
    tol = 1e-2 * max(1.0, float(np.abs(lo_ref).max()))
    err = float(np.abs(lo0[0].cpu().numpy() - lo_ref[0]).max())
    assert err <= tol, f"MoE context phase differs from the oracle by {err:.3e}"
    h = Host(model, batch, max_len, span, "none")
    assert h.report["fused"] and h.report["device_resident"] and h.report["layers"] == len(model.layers), h.report["why"]
    assert h.report["types"].count("DihipMoeBlock") == len(model.layers) and "MOEA16W8" not in h.report["types"]
    for b, pr in enumerate(prompts):
        k, v = h.spans()
        first = h.start(pr, k, v)
        assert first == ids0[b], f"request {b}: first id {first} != {ids0[b]}"
        assert torch.equal(h.logits()[0], lo0[b]), f"request {b}: context-phase logits are not bit-identical"
    for t in range(steps):
        ids = h.steps(1, graph=True)
        assert ids == want[t][1], f"step {t}: ids differ"
        assert torch.equal(h.logits(), want[t][0]), f"step {t}: logits are not bit-identical to DecodeSession"
    h.close()
    # the unfused list: every reference operator its own launch, 16-bit activations between them
    u = Host(model, batch, max_len, span, "none", fuse=False)
    assert not u.report["fused"] and "MOEA16W8" in u.report["types"] and "CalcExpert" in u.report["types"]
    for b, pr in enumerate(prompts):
        k, v = u.spans()
        u.start(pr, k, v)
    worst = 0.0
    for t in range(2):
        u.steps(1, graph=False)
        lo_u = u.logits().float()
        scale = max(1.0, float(want[t][0].abs().max()))
        worst = max(worst, float((lo_u - want[t][0]).abs().max()) / scale)
        if u.m.sync_ids() != want[t][1]:
            break   # a near-tie resolved differently: the histories diverge from here on
    assert worst <= 3e-2, f"unfused MoE list differs from the fused one by {worst:.3e} (relative to max |logit|)"
    print(f"MoE host runner: context vs oracle {err:.2e}; unfused vs fused {worst:.2e}")
    u.close()


def test_exported_graph_arities_run_fused_and_bit_identical(pkg):
    """the list with the converter's own arities (Rotary + position mask, attention + beam index, GenerateOp with two inputs and
    three outputs, UpdateId behind it) goes through the same fused operators: ids and logits bit-identical to DecodeSession"""
    from dash_infer_amd import decoder
    cfg = decoder.ModelConfig("runner-test-exported", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=4242, keep_fp=True)
    span, max_len, steps, batch = 16, 64, 4, 3
    rng = np.random.default_rng(12)
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in (5, 17, 30)]
    sess = decoder.DecodeSession(model, batch, max_len=max_len, span_len=span, kv_mode="none")
    lo0 = sess.prefill(prompts).clone()
    ids0 = sess.ids.cpu().tolist()
    want = []
    for _ in range(steps):
        sess.step()
        torch.cuda.synchronize()
        want.append((sess.logits.clone(), sess.ids.cpu().tolist()))
    h = Host(model, batch, max_len, span, "none", exported=True)
    assert h.report["fused"] and h.report["device_resident"] and "UpdateId" in h.report["why"], h.report["why"]
    for b, pr in enumerate(prompts):
        k, v = h.spans()
        assert h.start(pr, k, v) == ids0[b]
        assert torch.equal(h.logits()[0], lo0[b])
    for t in range(steps):
        assert h.steps(1, graph=True) == want[t][1]
        assert torch.equal(h.logits(), want[t][0]), f"step {t}"
    h.close()


def test_fused_operator_list_f16_is_bit_identical_to_decode_session(pkg):
    """the same with f16 activations (scales / zeros / embedding / lm_head in f16): the fused operators carry the weights' FT"""
    from dash_infer_amd import decoder
    cfg = decoder.ModelConfig("runner-test-f16", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=77, keep_fp=True, dtype=torch.float16)
    span, max_len, steps, batch = 16, 64, 4, 2
    rng = np.random.default_rng(2)
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in (9, 20)]
    sess = decoder.DecodeSession(model, batch, max_len=max_len, span_len=span, kv_mode="none")
    lo0 = sess.prefill(prompts).clone()
    ids0 = sess.ids.cpu().tolist()
    want = []
    for _ in range(steps):
        sess.step()
        torch.cuda.synchronize()
        want.append((sess.logits.clone(), sess.ids.cpu().tolist()))
    h = Host(model, batch, max_len, span, "none", ft="f16")
    assert h.report["fused"], h.report["why"]
    for b, pr in enumerate(prompts):
        k, v = h.spans()
        assert h.start(pr, k, v) == ids0[b]
        assert torch.equal(h.logits()[0], lo0[b])
    for t in range(steps):
        assert h.steps(1, graph=True) == want[t][1]
        assert torch.equal(h.logits(), want[t][0]), f"step {t}"
    h.close()


def test_eager_and_graph_replay_agree_and_unfused_list_is_close(pkg):
    from dash_infer_amd import decoder
    cfg = decoder.ModelConfig("runner-test", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=31, keep_fp=True)
    span, max_len, steps, B = 16, 64, 5, 2
    rng = np.random.default_rng(5)
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in (19, 7)]
    runs = {}
    for tag, fuse, graph in (("fused+graph", True, True), ("fused eager", True, False), ("unfused eager", False, False)):
        h = Host(model, B, max_len, span, "none", fuse=fuse)
        assert h.report["fused"] == fuse
        firsts = []
        for pr in prompts:
            k, v = h.spans()
            firsts.append(h.start(pr, k, v))
        out = []
        for _ in range(steps):
            ids = h.steps(1, graph=graph)
            out.append((h.logits().float().clone(), ids))
        runs[tag] = (firsts, out)
        h.close()
    assert runs["fused+graph"][0] == runs["fused eager"][0]
    for (la, ia), (lb, ib) in zip(runs["fused+graph"][1], runs["fused eager"][1]):
        assert ia == ib and torch.equal(la, lb)
    # the reference's own operator list (FT tensors between operators, fourteen launches per layer): same function, other rounding
    scale = max(1.0, float(runs["fused eager"][1][0][0].abs().max()))
    for t, ((lf, _), (lu, _)) in enumerate(zip(runs["fused eager"][1], runs["unfused eager"][1])):
        if t == 0 or runs["fused eager"][1][t - 1][1] == runs["unfused eager"][1][t - 1][1]:   # same inputs so far
            assert float((lf - lu).abs().max()) <= 3e-2 * scale, f"step {t}"
        else:
            break
    # graph replay of the UNFUSED list is refused (its operators stage host data per step)
    h = Host(model, B, max_len, span, "none", fuse=False)
    k, v = h.spans()
    h.start(prompts[0], k, v)
    from dash_infer_amd import hostapi
    with pytest.raises(hostapi.HostError) as e:
        h.steps(1, graph=True)
    assert "fused" in str(e.value)
    h.close()


@pytest.mark.parametrize("kv_mode", ["none", "i8"])
def test_requests_join_and_leave_the_running_batch(pkg, kv_mode):
    """Continuous batching at the operator level (model.cpp:1226-1246: Reshape on a membership change): A decodes alone, B joins
    after its own context phase, A leaves, C joins -- every request generates the tokens it generates in a batch of its own.
    A decisive synthetic model (decoder.build_random_model(decisive=...)): the ids do not hinge on near-ties."""
    from dash_infer_amd import decoder
    cfg = decoder.ModelConfig("runner-test", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=77, keep_fp=True, decisive=1.5)
    span, max_len = 16, 96
    rng = np.random.default_rng(9)
    prompts = {name: [int(t) for t in rng.integers(0, cfg.vocab, n)] for name, n in (("A", 11), ("B", 30), ("C", 5))}

    def alone(pr, n):
        sess = decoder.DecodeSession(model, 1, max_len=max_len, span_len=span, kv_mode=kv_mode)
        sess.prefill([pr])
        out = [int(sess.ids[0])]
        for _ in range(n):
            sess.step()
            out.append(int(sess.ids[0]))
        torch.cuda.synchronize()
        return out

    want = {name: alone(pr, 12) for name, pr in prompts.items()}
    h = Host(model, 3, max_len, span, kv_mode)
    got = {name: [] for name in prompts}
    order = []

    def start(name):
        k, v = h.spans()
        got[name].append(h.start(prompts[name], k, v))
        order.append(name)

    def run(n):
        for _ in range(n):
            ids = h.steps(1, graph=True)
            assert len(ids) == len(order)
            for name, i in zip(order, ids):
                got[name].append(i)

    start("A"); run(3)
    start("B"); run(3)                       # membership change: Reshape, state re-uploaded, step re-captured
    with torch.cuda.stream(h.stream):
        h.m.request_stop(order.index("A"))
    order.remove("A"); run(2)
    start("C"); run(4)
    h.close()
    for name in prompts:
        assert got[name] == want[name][: len(got[name])], f"request {name}: {got[name]} vs alone {want[name]}"
    assert len(got["A"]) == 7 and len(got["B"]) == 10 and len(got["C"]) == 5


@pytest.mark.parametrize("fuse", [False, True])
def test_prefill_over_a_cached_prefix_matches_a_prefill_from_scratch(pkg, fuse):
    """Prefix-cache hit through the whole operator list: request B shares its first 32 tokens (two spans) with request A, its
    cache starts on A's spans (prefix_len = 32, gen_ctx->step = prefix_len: model.cpp:532) and only the remaining tokens are
    prefilled -- rotated at positions step + t (rotary_op.cpp:331; ADVICE r3: the unfused Rotary operator counted the prefix
    twice).  Logits and next token must be those of the full prompt prefilled from scratch."""
    from dash_infer_amd import decoder
    cfg = decoder.ModelConfig("runner-test", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=12, keep_fp=True, decisive=1.5)
    span, max_len, P = 16, 96, 32
    rng = np.random.default_rng(3)
    shared = [int(t) for t in rng.integers(0, cfg.vocab, P)]
    tail_a = [int(t) for t in rng.integers(0, cfg.vocab, 9)]
    tail_b = [int(t) for t in rng.integers(0, cfg.vocab, 14)]
    h = Host(model, 3, max_len, span, "none", fuse=fuse)
    ka, va = h.spans()
    h.start(shared + tail_a, ka, va)
    # from scratch
    kf, vf = h.spans()
    first_full = h.start(shared + tail_b, kf, vf)
    lo_full = h.logits().float().clone()
    # over the cached prefix: A's first P / span spans, then its own
    kb, vb = h.spans()
    nsp = P // span
    kb = [ka[l][:nsp] + kb[l][nsp:] for l in range(len(ka))]
    vb = [va[l][:nsp] + vb[l][nsp:] for l in range(len(va))]
    first_pref = h.start(tail_b, kb, vb, prefix_len=P)
    lo_pref = h.logits().float().clone()
    scale = max(1.0, float(lo_full.abs().max()))
    err = float((lo_full - lo_pref).abs().max())
    assert err <= 1e-2 * scale, f"prefix-cache prefill differs from the prefill from scratch by {err:.3e} (scale {scale:.1f})"
    assert first_pref == first_full
    # and both continue identically
    ids = h.steps(3, graph=fuse)
    assert ids[1] == ids[2]
    h.close()


def test_layers_fused_behind_the_reference_tail(pkg):
    """A list whose tail the fused head does not cover (the tensor-parallel form: Gemm with splitk + AllReduce, here on one rank
    where the AllReduce copies) still gets its LAYERS fused, with the reference's own tail operators behind a DihipFinalNorm and -- since
    round 6 -- DihipGreedy as the sampling operator behind them, so the step state stays on the device and the step replays as a hipGraph
    (what the TP ranks run: tests/test_gpu_tp_loopback.py).  Logits (FT, from the Gemm operator) agree with the fully fused list to the
    FT rounding of the logits, eager and replayed."""
    from dash_infer_amd import decoder, hostapi, ops
    cfg = decoder.ModelConfig("runner-test", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=31, keep_fp=True)
    span, max_len, B = 16, 64, 2
    rng = np.random.default_rng(8)
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in (13, 6)]
    full = Host(model, B, max_len, span, "none")
    want = []
    for pr in prompts:
        k, v = full.spans()
        full.start(pr, k, v)
    for _ in range(3):
        ids = full.steps(1, graph=True)
        want.append((full.logits().float().clone(), ids))
    full.close()

    h = Host.__new__(Host)
    h.cfg, h.nl, h.spr = cfg, len(model.layers), (max_len + span - 1) // span
    h.stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    h.pool = ops.SpanPool(2 * B * h.nl * h.spr + 4, cfg.n_kv, span, cfg.head_dim, "none", torch.bfloat16)
    with torch.cuda.stream(h.stream):
        h.m = hostapi.Model(ops.cur_stream(), cfg.n_heads, cfg.n_kv, cfg.head_dim, span, 0, max_batch=B, max_len=max_len)
        ref_graph.register_weights(h.m, model)
        ref_graph.add_graph(h.m, ref_graph.qwen2_graph(h.nl, 4, 128, cfg.eps, cfg.n_heads, cfg.n_kv, cfg.rope_theta, tp_lm_head=True))
        h.report = h.m.graph_build(fuse=True)
    h.stream.synchronize()
    assert h.report["fused"] and h.report["device_resident"]
    assert h.report["types"][-5:] == ["DihipFinalNorm", "GetLastLine", "Gemm", "AllReduce", "DihipGreedy"]
    assert "DihipRopeSpanAttn" in h.report["types"] and "LayerNormNoBeta" not in h.report["types"]
    h.report["fused"] = False      # (Host.logits(): the tail's Gemm writes FT logits)
    for pr in prompts:
        k, v = h.spans()
        h.start(pr, k, v)
    scale = max(1.0, float(want[0][0].abs().max()))
    for t in range(3):
        ids = h.steps(1, graph=t > 0)   # the first step eager, the others from the captured graph
        lo = h.logits().float()
        assert float((lo - want[t][0]).abs().max()) <= 2 ** -7 * scale, f"step {t}"
        if ids != want[t][1]:
            break                   # a near-tie decided differently by the FT logits: later steps see other inputs
    h.close()


def test_a_graph_serialized_by_the_reference_converter_runs_bit_identically(pkg):
    """tests/golden/qwen2_a16w4_g128.asgraph.pb -- a TransformerProto whose graphs the reference's own Qwen_v15._build_graph + quantize_op
    wrote (tests/golden/make_graph_golden.py) -- through host/graph_wire.h, the fusion pass and the model runner: context phase and
    graph-replayed decode steps bit-identical to DecodeSession over the same weights (bound by the converter's names)."""
    import os
    from dash_infer_amd import decoder
    data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qwen2_a16w4_g128.asgraph.pb"), "rb").read()
    cfg = decoder.ModelConfig("export", hidden=3584, layers=2, n_heads=28, n_kv=4, head_dim=128, inter=1024, vocab=2048)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=77, keep_fp=True)
    span, max_len, steps = 128, 256, 4
    prompt = [int(t) for t in np.random.default_rng(3).integers(0, cfg.vocab, 70)]
    sess = decoder.DecodeSession(model, 1, max_len=max_len, span_len=span)
    lo0 = sess.prefill([prompt]).clone()
    want = []
    for _ in range(steps):
        sess.step()
        torch.cuda.synchronize()
        want.append((sess.logits.clone(), sess.ids.cpu().tolist()))
    h = Host(model, 1, max_len, span, "none", serialized=data)
    assert h.report["fused"] and h.report["layers"] == 2, h.report["why"]
    k, v = h.spans()
    first = h.start(prompt, k, v)
    assert first == int(lo0.argmax()) and torch.equal(h.logits()[0], lo0[0])
    for t in range(steps):
        ids = h.steps(1, graph=True)
        assert ids == want[t][1] and torch.equal(h.logits(), want[t][0]), f"step {t}"
    h.close()


def test_a_second_build_of_one_model_is_refused(pkg):
    """the weight-only operators release the model-owned source weights once re-laid-out, so a model is built ONCE -- a second
    dihost_graph_build (fuse = 0 after fuse = 1, a retry) is an INVALID_CALL with a message, not a pack from freed memory."""
    from dash_infer_amd import decoder, hostapi
    cfg = decoder.ModelConfig("runner-test", **SMALL)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=3, keep_fp=True)
    h = Host(model, 1, 64, 16, "none")
    assert h.report["fused"]
    with pytest.raises(hostapi.HostError, match="built already"):
        h.m.graph_build(fuse=False)
    k, v = h.spans()
    h.start([1, 2, 3], k, v)          # the first build still serves
    assert len(h.steps(2, graph=True)) == 1
    h.close()
