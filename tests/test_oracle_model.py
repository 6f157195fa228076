"""The whole-decoder oracle (oracle/model.py) checked against itself on CPU: token-by-token decoding against the growing
(possibly quantised) cache must give the same logits as recomputing the whole sequence with the prefill attention
oracle -- two evaluation orders of one function, built from pieces that are pinned separately against the reference."""
import numpy as np
import pytest

from oracle import glue, model as omodel, quant
from oracle.numerics import bf16_round


def make_oracle(rng, wbits, group, kv_mode, hidden=256, n=2, g=1, H=128, inter=256, vocab=64, nlayers=2):
    def qlin(K, N):
        W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
        return (quant.iq_quantize_a16w8 if wbits == 8 else quant.iq_quantize_a16w4)(W, group, "bf16")
    layers = []
    for _ in range(nlayers):
        layers.append(dict(qkv=qlin(hidden, (n + 2 * g) * H), o=qlin(n * H, hidden), gate=qlin(hidden, inter), up=qlin(hidden, inter),
                           down=qlin(inter, hidden), qkv_bias=bf16_round(rng.normal(0, 0.05, (n + 2 * g) * H).astype(np.float32)),
                           ln1=bf16_round(1 + rng.normal(0, 0.1, hidden).astype(np.float32)),
                           ln2=bf16_round(1 + rng.normal(0, 0.1, hidden).astype(np.float32))))
    embed = bf16_round(rng.normal(0, 1.0, (vocab, hidden)).astype(np.float32))
    fn = bf16_round(1 + rng.normal(0, 0.1, hidden).astype(np.float32))
    lm = bf16_round(rng.normal(0, 0.05, (hidden, vocab)).astype(np.float32))
    return omodel.DecoderOracle(layers, embed, fn, lm, n, g, H, wbits, group, kv_mode=kv_mode)


@pytest.mark.parametrize("wbits,group,kv_mode", [(8, -1, "none"), (4, 128, "none"), (4, 128, "i8"), (8, 128, "u4")])
def test_incremental_decode_equals_full_recompute(wbits, group, kv_mode):
    rng = np.random.default_rng(wbits + len(kv_mode))
    m = make_oracle(rng, wbits, group, kv_mode)
    seq = rng.integers(0, 64, 7)
    step_logits = [m.step([t])[0] for t in seq]
    for L in (1, 4, 7):
        full = m.last_logits_from_scratch(seq[:L])[0]
        # same operands and rounding points; only the softmax / P.V summation order of the two attention oracles differs
        np.testing.assert_allclose(step_logits[L - 1], full, rtol=0, atol=2e-5 * max(1.0, float(np.abs(full).max())))


def test_batch_rows_are_independent():
    rng = np.random.default_rng(3)
    m1, m2 = make_oracle(np.random.default_rng(5), 4, 128, "u4"), make_oracle(np.random.default_rng(5), 4, 128, "u4")
    ids = rng.integers(0, 64, (3, 4))                      # 3 steps, batch 4
    for t in range(3):
        both = m1.step(ids[t])
        one = m2.step(ids[t][1:2])
        np.testing.assert_array_equal(both[1], one[0])


@pytest.mark.parametrize("rounding", ["x86", "ft_graph"])
@pytest.mark.parametrize("kv_mode", ["none", "i8"])
def test_prefill_then_step_equals_step_by_step(rounding, kv_mode):
    """DecoderOracle.prefill (whole prompt at once, causal prefill attention over the fresh K / V, weights dequantised once)
    leaves the cache token-by-token decoding would have left: with the 16-bit cache the two evaluation orders give the
    same logits to summation-order accuracy and later steps continue identically; both rounding modes."""
    rng = np.random.default_rng(11)
    a = make_oracle(rng, 4, 128, kv_mode)
    b = make_oracle(np.random.default_rng(11), 4, 128, kv_mode)
    a.rounding = b.rounding = rounding
    a._wcache = {}
    seqs = [[int(t) for t in rng.integers(0, 64, 9)], [int(t) for t in rng.integers(0, 64, 5)]]
    lo = a.prefill(seqs)
    for bi, seq in enumerate(seqs):
        c = make_oracle(np.random.default_rng(11), 4, 128, kv_mode)
        c.rounding = rounding
        for t in seq:
            ref = c.step([t])[0]
        if kv_mode == "none":
            np.testing.assert_allclose(lo[bi], ref, rtol=0, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
        else:
            # the context phase attends over the fresh (unquantised) K / V, step() over the quantised cache (as the reference)
            np.testing.assert_allclose(lo[bi], ref, rtol=0, atol=0.15 * max(1.0, float(np.abs(ref).max())))
    nxt = a.step([3, 4])
    assert nxt.shape == (2, 64) and np.isfinite(nxt).all()
    assert len(a.cache[0][0][0]) == 10 and len(a.cache[0][1][0]) == 6


def test_rounding_modes_differ_only_at_the_documented_points():
    rng = np.random.default_rng(4)
    a = make_oracle(rng, 4, 128, "none")
    b = make_oracle(np.random.default_rng(4), 4, 128, "none")
    b.rounding = "ft_graph"
    seq = [1, 5, 9]
    for t in seq:
        la, lb = a.step([t])[0], b.step([t])[0]
    d = float(np.abs(la - lb).max())
    assert 0 < d < 0.1 * max(1.0, float(np.abs(la).max())), d  # bf16 residual / gate / up rounding: small but not zero


def add_moe(rng, m, wbits, group, hidden=256, num_experts=4, top_k=2, moe_inter=128):
    def qlin(K, N):
        W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
        return (quant.iq_quantize_a16w8 if wbits == 8 else quant.iq_quantize_a16w4)(W, group, "bf16")
    for lw in m.layers:
        lw["moe"] = {"router": bf16_round(rng.normal(0, 0.5, (hidden, num_experts)).astype(np.float32)), "top_k": top_k,
                     "shared_gate_w": bf16_round(rng.normal(0, 0.05, (hidden, 1)).astype(np.float32)),
                     "experts_gate": [qlin(hidden, moe_inter) for _ in range(num_experts)],
                     "experts_up": [qlin(hidden, moe_inter) for _ in range(num_experts)],
                     "experts_down": [qlin(moe_inter, hidden) for _ in range(num_experts)]}
    return m


def test_moe_layers_incremental_equals_full_recompute_and_route_through_the_experts():
    """the mixture-of-experts block (qwen_v20_moe.py:318-382) in both evaluation orders; and it IS the sum of its parts:
    with the routed experts' down projections zeroed the block reduces to the gated shared expert, with the shared expert's
    zeroed to the routed experts alone"""
    rng = np.random.default_rng(11)
    m = add_moe(rng, make_oracle(rng, 8, -1, "none"), 8, -1)
    seq = rng.integers(0, 64, 5)
    step_logits = [m.step([t])[0] for t in seq]
    full = m.last_logits_from_scratch(seq)[0]
    np.testing.assert_allclose(step_logits[-1], full, rtol=0, atol=2e-5 * max(1.0, float(np.abs(full).max())))
    # parts: zero the shared expert's down projection -> the routed experts alone; zero the routed experts' -> the gated shared expert
    h = rng.normal(0, 1, (3, 256)).astype(np.float32)
    lw = m.layers[0]
    zeroed = lambda w: (w[0], w[1] * 0, w[2])
    routed_only = dict(lw, down=zeroed(lw["down"]))
    shared_only = dict(lw, moe=dict(lw["moe"], experts_down=[zeroed(w) for w in lw["moe"]["experts_down"]]))
    whole, a, b = m._moe_mlp(h, lw) - h, m._moe_mlp(h, routed_only) - h, m._moe_mlp(h, shared_only) - h
    np.testing.assert_allclose(whole, a + b, rtol=0, atol=1e-5)
    assert np.abs(a).max() > 1e-3 and np.abs(b).max() > 1e-3


@pytest.mark.parametrize("rounding", ["x86", "ft_graph", "x86_pure_bf16", "x86_pure_bf16_exactw", "x86_pure_f32"])
def test_teacher_forced_pass_equals_prefill_then_steps(rounding):
    """oracle.model.teacher_forced_logits (what the full-depth GPU comparison uses: one pass over the weights, all positions
    at once, several roundings in lockstep) gives the logits of prefill + step() fed the same tokens, for every rounding
    mode -- including the pure x86 ones (f32 qkv / cache / attention output; bf16 weight reorder under medium_bf16)."""
    rng = np.random.default_rng(23)
    a = make_oracle(rng, 4, 128, "none")
    a.rounding = rounding
    seq = [int(t) for t in rng.integers(0, 64, 11)]
    L, n_last = 7, 5
    ref = [a.prefill([seq[:L]])[0]]
    for t in range(L, L + n_last - 1):
        ref.append(a.step([seq[t]])[0])
    others = [make_oracle(np.random.default_rng(23), 4, 128, "none") for _ in range(2)]
    others[0].rounding, others[1].rounding = rounding, "x86_pure_f32"
    got = omodel.teacher_forced_logits(others, others[0].layers, seq[:L + n_last - 1], n_last)
    np.testing.assert_allclose(got[0], np.stack(ref), rtol=0, atol=2e-5 * max(1.0, float(np.abs(np.stack(ref)).max())))
    # f32 accumulation (the full-depth setting): the same function up to the summation order -- which, through bf16
    # roundings that flip, already moves a logit by several 1e-3 on this two-layer model (the noise floor every
    # "logits within 1e-2" statement about a bf16 graph sits on)
    for o in others:
        o.acc = np.float32
    got32 = omodel.teacher_forced_logits(others, others[0].layers, seq[:L + n_last - 1], n_last)
    np.testing.assert_allclose(got32[0], got[0], rtol=0, atol=1e-2 * max(1.0, float(np.abs(got[0]).max())))


def test_pure_x86_rounding_modes_differ_where_they_should():
    """The rounding specs are not aliases: the bf16 weight reorder, the FT cache and the f32-everything path give different
    logits on the same model, at the size of the roundings they add / remove (sanity for the ablation in DESIGN section 0)."""
    rng = np.random.default_rng(29)
    ms = {}
    for r in ("x86", "x86_pure_bf16", "x86_pure_bf16_exactw", "x86_pure_f32"):
        ms[r] = make_oracle(np.random.default_rng(29), 4, 128, "none")
        ms[r].rounding = r
    seq = [int(t) for t in rng.integers(0, 64, 9)]
    lo = dict(zip(ms, omodel.teacher_forced_logits(list(ms.values()), ms["x86"].layers, seq, 1)))
    d = lambda a, b: float(np.abs(lo[a] - lo[b]).max())
    assert 0 < d("x86", "x86_pure_bf16_exactw") < 5e-2       # FT rounding of qkv / cache / attention output
    assert 0 < d("x86_pure_bf16", "x86_pure_bf16_exactw") < 5e-2   # weight rounding
    assert 0 < d("x86_pure_f32", "x86_pure_bf16_exactw") < 5e-2    # src rounding


# BASELINE configs[0]: "Qwen2-0.5B bf16 greedy decode, batch 1, x86 CPU reference path (plumbing, no GPU)".  The reference's
# CPU engine entry is switched off in this version and its x86 libraries are LFS stubs (F4): the path exists
# here as the oracle's unquantised form -- op type Gemm with bf16 weights (gemm_op_cpu.cpp:75-126), head size 64, 14 query /
# 2 KV heads, f32 tensors between operators.  Qwen2-0.5B widths, a few layers, a small vocabulary (CPU seconds).
def make_dense_oracle(rng, rounding, hidden=896, n=14, g=2, H=64, inter=4864, vocab=512, nlayers=3):
    w = lambda K, N, std=0.02: bf16_round(rng.normal(0, std, (K, N)).astype(np.float32))
    layers = [dict(qkv=w(hidden, (n + 2 * g) * H), o=w(n * H, hidden), gate=w(hidden, inter), up=w(hidden, inter), down=w(inter, hidden),
                   qkv_bias=bf16_round(rng.normal(0, 0.05, (n + 2 * g) * H).astype(np.float32)),
                   ln1=bf16_round(1 + rng.normal(0, 0.1, hidden).astype(np.float32)),
                   ln2=bf16_round(1 + rng.normal(0, 0.1, hidden).astype(np.float32))) for _ in range(nlayers)]
    embed = bf16_round(rng.normal(0, 1.0, (vocab, hidden)).astype(np.float32))
    fn = bf16_round(1 + rng.normal(0, 0.1, hidden).astype(np.float32))
    lm = w(hidden, vocab, 0.05)
    return omodel.DecoderOracle(layers, embed, fn, lm, n, g, H, 16, -1, kv_mode="none", rounding=rounding)


@pytest.mark.parametrize("rounding", ["x86_pure_bf16", "x86_pure_f32"])
def test_config0_unquantised_bf16_model_incremental_decode_equals_full_recompute(rounding):
    """configs[0] plumbing: greedy decoding of an unquantised Qwen2-0.5B-shaped model on the x86 semantics (medium_bf16 and the
    default f32 matmul precision), token by token against the growing f32 cache == recomputing the whole sequence (prefill
    attention oracle) == the one-pass teacher-forced evaluation; greedy ids identical along all three."""
    rng = np.random.default_rng(64)
    m = make_dense_oracle(rng, rounding)
    prompt = [int(t) for t in rng.integers(0, 512, 6)]
    lo = m.prefill([prompt])
    seq, step_logits = list(prompt), [lo[0]]
    for _ in range(5):
        nxt = int(glue.greedy(step_logits[-1][None, :])[0])
        seq.append(nxt)
        step_logits.append(m.step([nxt])[0])
    m2 = make_dense_oracle(np.random.default_rng(64), rounding)
    for L in (len(prompt), len(prompt) + 2, len(seq)):
        full = m2.last_logits_from_scratch(seq[:L])[0]
        want = step_logits[L - len(prompt)]
        np.testing.assert_allclose(want, full, rtol=0, atol=3e-5 * max(1.0, float(np.abs(full).max())))
        assert int(np.argmax(want)) == int(np.argmax(full))
    tf = omodel.teacher_forced_logits([m2], m2.layers, seq, len(step_logits))[0]
    np.testing.assert_allclose(tf, np.stack(step_logits), rtol=0, atol=3e-5 * max(1.0, float(np.abs(tf).max())))


@pytest.mark.parametrize("kv_mode,rounding", [("none", "x86"), ("u4", "x86"), ("i8", "ft_graph")])
def test_layer_major_trace_equals_prefill_then_steps(kv_mode, rounding):
    """teacher_forced_trace (every layer applied to all rows of all requests before the next layer is touched; context rows
    attend over the fresh K / V, decoder rows over the cache's codec image) is the same function as prefill() followed by
    step() token by token -- for ragged batches and the quantised caches too.  Its per-layer hook sees each layer's inputs,
    outputs and the cache image of K / V (what the per-layer drift test of tests/test_gpu_parity_depth.py feeds the GPU)."""
    rng = np.random.default_rng(21)
    a = make_oracle(rng, 4, 128, kv_mode, nlayers=3)
    b = make_oracle(np.random.default_rng(21), 4, 128, kv_mode, nlayers=3)
    a.rounding = b.rounding = rounding
    prompts = [[int(t) for t in rng.integers(0, 64, n)] for n in (6, 3)]
    fed = [[int(t) for t in rng.integers(0, 64, 4)] for _ in prompts]       # the tokens fed at 4 decode steps
    want = [b.prefill(prompts)]
    for t in range(4):
        want.append(b.step([f[t] for f in fed]))
    seen = []
    def hook(li, h_in, h_out, kvs):
        seen.append(li)
        assert [h.shape for h in h_in] == [h.shape for h in h_out] == [(10, 256), (7, 256)]
        assert kvs[0][0].shape == (10, 1, 128) and kvs[1][1].shape == (7, 1, 128)
        if li == 0:   # the cache image of the trace is what step() stored
            np.testing.assert_array_equal(kvs[0][2][7], b.cache[0][0][0][7])
            np.testing.assert_array_equal(kvs[1][3][2], b.cache[0][1][1][2])
            if kv_mode == "none":
                assert kvs[0][0] is kvs[0][2]
            else:   # the written rows are FT-valued, their image is the codec's
                np.testing.assert_array_equal(kvs[0][0], bf16_round(kvs[0][0]))
                assert np.abs(kvs[0][0] - kvs[0][2]).max() > 0
    got = omodel.teacher_forced_trace(a, a.layers, [p + f for p, f in zip(prompts, fed)], [6, 3], on_layer=hook, want_kv=True)
    assert seen == [0, 1, 2]
    for bi in range(2):
        assert got[bi].shape == (5, 64)
        for t in range(5):
            ref = want[t][bi]
            np.testing.assert_allclose(got[bi][t], ref, rtol=0, atol=3e-5 * max(1.0, float(np.abs(ref).max())))
