"""The UNMODIFIED reference layer graph on DeviceType::HIP (first slice).

The operator list of python/pyhie/allspark/model/qwen_v15.py:187-388 + model_base.py:690-703 in its quantised form (dynamic
quantisation switches the fused binary ADD off, qwen_v15.py:175-178) is built from OperatorProto structs, every op type is
looked up in the OpFactory for DeviceType::HIP (what AsModel does, csrc/core/model/model.cpp:265-287 -- an unregistered type
throws "Unsupported op type.", operator.cpp:379-386) and driven CallInit -> CallReshape -> CallAlloc -> CallForward in graph
order, context phase then decoder phase, tensors bound by name in one TensorMap:

  EmbeddingT5 -> L x [ LayerNormNoBeta -> GemmA16W4/W8 (qkv, bias) -> Rotary -> DecOptMQA -> GemmA16Wx (o) -> Binary ADD ->
                       LayerNormNoBeta -> GemmA16Wx (gate, SILU) | GemmA16Wx (up) -> Binary MUL -> GemmA16Wx (down) -> Binary ADD ]
              -> LayerNormNoBeta -> GetLastLine -> Gemm (lm_head) -> GenerateOp (greedy)

Every tensor between operators is an FT tensor, as on the reference's GPU path: the comparison target is the oracle's
`ft_graph` rounding (oracle/model.py); the product's fused decode step (decoder.DecodeSession: f32 hidden stream, the `x86`
rounding points) must agree with the graph to the distance of those rounding points."""
import numpy as np
import pytest
import torch

from oracle import glue, model as omodel

pytestmark = pytest.mark.gpu


def view_of(ptr, shape, dtype):
    import ctypes as C
    n = int(np.prod(shape))
    t = torch.empty(n, dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), n * t.element_size(), 3) == 0
    return t.view(*shape)


def build_graph(m, model, wbits, group):
    """-> op ids in graph order; weights registered under the reference's names."""
    cfg = model.cfg
    fp = model.fp
    gemm_type = "GemmA16W4" if wbits == 4 else "GemmA16W8"
    gattr = f"GroupSize=i:{group}" if group > 0 else ""
    qdt = "u8" if wbits == 4 else "i8"
    ops_ = []

    def lowp(name, key, li, inputs, out, act=0, bias=None):
        q, s, z = fp[li][key]
        m.set_weight(name + ".weight", q, qdt)
        m.set_weight(name + ".weight.scale", s, "bf16")
        m.set_weight(name + ".weight.zero_point", z, "bf16")
        w = [name + ".weight", name + ".weight.scale", name + ".weight.zero_point"]
        if bias is not None:
            m.set_weight(name + ".bias", bias, "bf16")
            w.append(name + ".bias")
        attrs = ";".join(a for a in (gattr, f"activation=i:{act}" if act else "", "alpha=f:1.0") if a)
        ops_.append(m.create_op(gemm_type, name, inputs, [out], w, attrs))

    m.set_weight("embedding.word_embeddings", fp["embed"], "bf16")
    ops_.append(m.create_op("EmbeddingT5", "embedding", ["input_ids"], ["embedding.out"], ["embedding.word_embeddings"], "token_embedding=b:0"))
    prev = "embedding.out"
    for li in range(len(model.layers)):
        p = f"decoder.layer.{li}."
        m.set_weight(p + "attention.layernorm.gamma", fp[li]["ln1"], "bf16")
        m.set_weight(p + "ffn.layernorm.gamma", fp[li]["ln2"], "bf16")
        ops_.append(m.create_op("LayerNormNoBeta", p + "attention.layernorm", [prev], [p + "attention.layernorm.out"],
                                [p + "attention.layernorm.gamma"], f"eps=f:{cfg.eps}"))
        lowp(p + "attention.self", "qkv", li, [p + "attention.layernorm.out"], p + "attention.self.out", bias=fp[li]["qkv_bias"])
        ops_.append(m.create_op("Rotary", p + "rotary", [p + "attention.self.out"], [p + "rotary.out"], [],
                                f"num_heads=i:{cfg.n_heads};multi_query_group_num=i:{cfg.n_kv};rotary_base=f:{cfg.rope_theta}"))
        ops_.append(m.create_op("DecOptMQA", p + "attention", [p + "rotary.out"], [p + "attention.out"]))
        lowp(p + "attention.output.dense", "o", li, [p + "attention.out"], p + "attention.output.dense.out")
        ops_.append(m.create_op("Binary", p + "attention_add", [p + "attention.output.dense.out", prev], [p + "attention_add.out"], [], "binary_type=i:1"))
        ops_.append(m.create_op("LayerNormNoBeta", p + "ffn.layernorm", [p + "attention_add.out"], [p + "ffn.layernorm.out"],
                                [p + "ffn.layernorm.gamma"], f"eps=f:{cfg.eps}"))
        lowp(p + "ffn.intermediate.dense", "gate", li, [p + "ffn.layernorm.out"], p + "ffn.intermediate.dense.out", act=5)
        lowp(p + "ffn.linear.dense", "up", li, [p + "ffn.layernorm.out"], p + "ffn.linear.dense.out")
        ops_.append(m.create_op("Binary", p + "ffn.mul", [p + "ffn.intermediate.dense.out", p + "ffn.linear.dense.out"], [p + "ffn.mul.out"], [], "binary_type=i:2"))
        lowp(p + "ffn.output.dense", "down", li, [p + "ffn.mul.out"], p + "ffn.output.dense.out")
        ops_.append(m.create_op("Binary", p + "final_add", [p + "ffn.output.dense.out", p + "attention_add.out"], [p + "final_add.out"], [], "binary_type=i:1"))
        prev = p + "final_add.out"
    m.set_weight("final.layernorm.gamma", fp["final_norm"], "bf16")
    m.set_weight("lm_head.weight", fp["lm_head"], "bf16")
    ops_.append(m.create_op("LayerNormNoBeta", "final.layernorm", [prev], ["last_hidden_state"], ["final.layernorm.gamma"], f"eps=f:{cfg.eps}"))
    ops_.append(m.create_op("GetLastLine", "get_last_line", ["last_hidden_state"], ["get_last_line.out"]))
    ops_.append(m.create_op("Gemm", "lm_head", ["get_last_line.out"], ["logits"], ["lm_head.weight"], "with_bias=b:0"))
    ops_.append(m.create_op("GenerateOp", "generate", ["logits"], ["generated_ids"], [], "top_k=i:1"))
    return ops_


def run_graph(m, ops_):
    for o in ops_:
        m.reshape(o)
    for o in ops_:
        m.alloc(o)
    for o in ops_:
        m.forward(o)
    torch.cuda.synchronize()
    _, shp, ptr = m.get_tensor("logits")
    logits = view_of(ptr, shp, torch.bfloat16).float().cpu().numpy().reshape(-1, shp[-1])
    _, shp2, ptr2 = m.get_tensor("generated_ids")
    ids = view_of(ptr2, shp2, torch.int64).cpu().numpy().reshape(-1)
    return logits, ids


@pytest.mark.parametrize("wbits,group,kv_mode", [(4, 128, "none"), (8, -1, "i8")])
def test_reference_layer_graph_resolves_and_runs_on_hip(pkg, wbits, group, kv_mode):
    from dash_infer_amd import decoder, hostapi, ops
    from tests.test_gpu_decoder import oracle_of
    registered = hostapi.lib().dihost_registered_ops().decode().split(",")
    for t in ("EmbeddingT5", "LayerNormNoBeta", "GemmA16W4", "GemmA16W8", "Rotary", "DecOptMQA", "Binary", "Unary", "UnaryGLU", "Gemm",
              "GetLastLine", "GenerateOp", "AllReduce", "AllGather"):
        assert t in registered, f"{t} is not registered for DeviceType::HIP"
    cfg = decoder.ModelConfig("graph-test", hidden=512, layers=2, n_heads=4, n_kv=2, head_dim=128, inter=1024, vocab=2048)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(wbits, group), seed=515, keep_fp=True)
    B, S, steps = 2, 16, 4
    max_len = 64
    spr = max_len // S
    nl = len(model.layers)
    rng = np.random.default_rng(wbits)
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in (21, 9)]
    pool = ops.SpanPool(2 * B * nl * spr + 1, cfg.n_kv, S, cfg.head_dim, kv_mode, torch.bfloat16)
    kspans = [[[pool.alloc()[0] for _ in range(spr)] for _ in range(nl)] for _ in range(B)]
    vspans = [[[pool.alloc()[0] for _ in range(spr)] for _ in range(nl)] for _ in range(B)]
    m = hostapi.Model(ops.cur_stream(), cfg.n_heads, cfg.n_kv, cfg.head_dim, S, {"none": 0, "i8": 1, "u4": 2}[kv_mode], max_batch=B, max_len=max_len)
    graph = None
    # ---- context phase, one request at a time (the reference prefills a request per pass)
    g_logits0, g_ids = [], []
    for b, pr in enumerate(prompts):
        ids_t = torch.tensor([pr], dtype=torch.int64, device="cuda")
        m.set_tensor("input_ids", ids_t, "i64")
        if graph is None:
            graph = build_graph(m, model, wbits, group)
        m.set_runtime(True, [0], [kspans[b]], [vspans[b]])
        lo, ids = run_graph(m, graph)
        g_logits0.append(lo[0])
        g_ids.append(int(ids[0]))
    # ---- decoder phase, the batch together
    cur = list(g_ids)
    lens = [len(p) for p in prompts]
    g_steps = []
    for t in range(steps):
        m.set_tensor("input_ids", torch.tensor([[c] for c in cur], dtype=torch.int64, device="cuda"), "i64")
        m.set_runtime(False, [l + t for l in lens], kspans, vspans)
        lo, ids = run_graph(m, graph)
        g_steps.append((lo, ids.copy()))
        cur = [int(i) for i in ids]
    m.close()

    # ---- oracle, ft_graph rounding (every operator output an FT tensor)
    ref = oracle_of(model, kv_mode)
    ref.rounding = "ft_graph"
    lo0 = ref.prefill(prompts)
    tol_unit = 1.5e-2   # the lm_head Gemm's FT (bf16) logits add half an ulp of |logit| to the graph's own roundings
    worst, decided = 0.0, 0
    def check(got, want, got_ids, tag):
        nonlocal worst, decided
        tol = tol_unit * max(1.0, float(np.abs(want).max()))
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        assert err <= tol, f"{tag}: logits differ by {err:.3e} (max |logit| {np.abs(want).max():.2f})"
        top2 = np.sort(want, axis=-1)[:, -2:]
        sure = (top2[:, 1] - top2[:, 0]) > 2 * tol
        assert np.array_equal(np.asarray(got_ids)[sure], glue.greedy(want)[sure]), f"{tag}: greedy ids differ"
        decided += int(sure.sum())
    check(np.stack(g_logits0), lo0, g_ids, "context")
    feed = list(g_ids)
    for t, (lo, ids) in enumerate(g_steps):
        want = ref.step(feed)
        check(lo, want, ids, f"decode step {t}")
        feed = [int(i) for i in ids]
    assert decided >= 1   # (a random 2-layer model has small top-2 margins: most positions are near-ties at this tolerance)

    # ---- the product's fused decode step on the same model: same function, other rounding points (f32 hidden stream)
    sess = decoder.DecodeSession(model, B, max_len=max_len, span_len=S, kv_mode=kv_mode)
    lo_fused0 = sess.prefill(prompts).cpu().numpy()
    scale = max(1.0, float(np.abs(lo0).max()))
    d0 = float(np.abs(lo_fused0 - np.stack(g_logits0)).max())
    assert d0 <= 3e-2 * scale, f"fused decode step vs operator graph (context): {d0:.3e}"
    sess.set_state(g_ids, lens)
    for t, (lo, ids) in enumerate(g_steps):
        sess.step()
        torch.cuda.synchronize()
        d = float(np.abs(sess.logits.cpu().numpy() - lo).max())
        assert d <= 3e-2 * scale, f"fused decode step vs operator graph (step {t}): {d:.3e}"
        sess.set_state(ids, [l + t + 1 for l in lens])   # follow the graph's tokens
    print(f"operator graph vs ft_graph oracle: worst logit error {worst:.2e}; {decided} decisive greedy choices; "
          f"vs the fused decode step (context) {d0:.2e}")


def test_unregistered_op_type_is_refused_like_the_reference(pkg):
    from dash_infer_amd import hostapi, ops
    m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16)
    with pytest.raises(hostapi.HostError) as e:
        m.create_op("NoSuchOp", "x", ["a"], ["b"])
    assert "Unsupported op type." in str(e.value)
    m.close()


@pytest.mark.parametrize("nranks", [1, 2])
def test_dense_gemm_operator_with_splitk_attribute(pkg, nranks):
    """op type Gemm on HIP: bias, activation, fused binary ADD, and the K-split form of the TP lm_head (attribute splitk,
    gemm_op.cpp:95-98: the input row keeps its full width, rank r multiplies columns [r k, (r + 1) k) with its row block) --
    the partial products of the ranks sum to the full product."""
    from dash_infer_amd import hostapi, ops
    from oracle.numerics import bf16_round
    rng = np.random.default_rng(7 + nranks)
    K, N, M = 256, 320, 3
    W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
    x = bf16_round(rng.normal(0, 1, (M, 1, K)).astype(np.float32))
    bias = bf16_round(rng.normal(0, 0.2, N).astype(np.float32))
    res = bf16_round(rng.normal(0, 1, (M, 1, N)).astype(np.float32))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).cuda()
    if nranks == 1:
        m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16)
        m.set_weight("w", dev(W), "bf16")
        m.set_weight("b", dev(bias), "bf16")
        m.set_tensor("x", dev(x), "bf16")
        m.set_tensor("r", dev(res), "bf16")
        op = m.create_op("Gemm", "g", ["x", "r"], ["y"], ["w", "b"], "activation=i:5;binary_type=i:1;alpha=f:0.5")
        m.reshape(op); m.alloc(op); m.forward(op)
        _, shp, ptr = m.get_tensor("y")
        assert shp == [M, 1, N]
        y = view_of(ptr, shp, torch.bfloat16).float().cpu().numpy().reshape(M, N)
        v = 0.5 * (x.reshape(M, K).astype(np.float64) @ W.astype(np.float64)) + bias
        want = bf16_round((v / (1 + np.exp(-v))).astype(np.float32)) + res.reshape(M, N)
        np.testing.assert_allclose(y, want, rtol=2 ** -7, atol=2 ** -7 * np.abs(want).max())
        m.close()
        return
    kloc = K // nranks
    total = np.zeros((M, N), np.float64)
    for r in range(nranks):
        m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16, rank=r, nranks=nranks)
        m.set_weight("w", dev(W[r * kloc:(r + 1) * kloc]), "bf16")
        m.set_tensor("x", dev(x), "bf16")      # the FULL row: lda = k * nranks
        op = m.create_op("Gemm", "lm_head", ["x"], ["y"], ["w"], "splitk=b:1")
        m.reshape(op); m.alloc(op); m.forward(op)
        _, shp, ptr = m.get_tensor("y")
        total += view_of(ptr, shp, torch.bfloat16).float().cpu().numpy().reshape(M, N)
        m.close()
    want = x.reshape(M, K).astype(np.float64) @ W.astype(np.float64)
    np.testing.assert_allclose(total, want, rtol=0, atol=nranks * 2 ** -8 * np.abs(want).max())
