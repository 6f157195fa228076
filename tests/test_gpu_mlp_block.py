"""dihip_decode_mlp_block -- RMSNorm + gate / up GEMV + SwiGLU and the down projection + residual of a batch-1 decode layer in ONE
launch -- against the two launches it replaces (dihip_fused_norm_swiglu, dihip_fused_gemm_addto): the hidden row must be
BIT-IDENTICAL (both phases are the stand-alone kernels' bodies: csrc/decode_mlp_block.hip), eager, back to back (own epoch in the
sync buffer) and under hipGraph replay; with dihip_decode_attn_block a decode layer is two launches.  qwen_v15.py:300-388."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _session(decoder, model, max_len, mlp, attn=None):
    keys = {"DIHIP_DECODER_MLP_BLOCK": "1" if mlp else "0"}
    if attn is not None:
        keys["DIHIP_DECODER_ATTN_BLOCK"] = "1" if attn else "0"
    old = {k: os.environ.get(k) for k in keys}
    os.environ.update(keys)
    try:
        return decoder.DecodeSession(model, 1, max_len=max_len, span_len=128)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("hidden,inter", [(3584, 18944), (3584, 2432), (1024, 4096)])
def test_one_launch_equals_the_two_it_replaces(pkg, hidden, inter):
    from dash_infer_amd import decoder, ops
    cfg = decoder.ModelConfig("mlp-block", hidden=hidden, layers=1, n_heads=hidden // 128, n_kv=max(1, hidden // 896), head_dim=128, inter=inter, vocab=1024)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128, gptq_like_zeros=True), seed=5)
    lw = model.layers[0]
    assert ops.decode_mlp_block_supported(lw.gate, hidden, torch.bfloat16, 1)
    sc = ops.Scratch(max(ops.lowp_workspace_bytes(4, 1, p.N, p.K, 128) for p in (lw.gate, lw.down)))
    sync = torch.zeros(int(ops.lib().dihip_decode_mlp_block_sync_bytes(inter)), dtype=torch.uint8, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(hidden + inter)
    for step in range(4):
        h = torch.randn(1, hidden, generator=gen, device="cuda", dtype=torch.float32) * (1.0 + step)
        act = ops.fused_norm_swiglu(h, lw.ln2, cfg.eps, lw.gate, lw.up, sc)
        want = ops.fused_gemm_addto(act, lw.down, h, sc, M=1)
        got = ops.decode_mlp_block(h, h, lw.ln2, cfg.eps, lw.gate, lw.up, lw.down, sync)
        nores = ops.decode_mlp_block(h, None, lw.ln2, cfg.eps, lw.gate, lw.up, lw.down, sync)   # row-parallel TP ranks > 0
        torch.cuda.synchronize()
        assert int(sync.view(torch.int32)[1].item()) == 0, "a bounded wait of the fused launch gave up"
        assert torch.equal(got, want), f"step {step}: max diff {(got - want).abs().max().item():.3e}"
        assert torch.equal(nores, ops.fused_gemm_addto(act, lw.down, None, sc, M=1))
    hh = torch.randn(1, hidden, generator=gen, device="cuda", dtype=torch.float32)
    out = torch.empty_like(hh)
    ops.decode_mlp_block(hh, hh, lw.ln2, cfg.eps, lw.gate, lw.up, lw.down, sync, out=out)   # warm (LDS grant) before the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.decode_mlp_block(hh, hh, lw.ln2, cfg.eps, lw.gate, lw.up, lw.down, sync, out=out)
    want = ops.fused_gemm_addto(ops.fused_norm_swiglu(hh, lw.ln2, cfg.eps, lw.gate, lw.up, sc), lw.down, hh, sc, M=1)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)


def test_two_launches_per_layer_decode_steps_are_bit_identical_to_the_chain(pkg):
    from dash_infer_amd import decoder
    cfg = decoder.ModelConfig("two-launch", hidden=3584, layers=2, n_heads=28, n_kv=4, head_dim=128, inter=2432, vocab=2048)
    model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128, gptq_like_zeros=True), seed=29)
    outs = []
    for fused in (False, True):
        s = _session(decoder, model, 512, mlp=fused, attn=fused)
        assert s.mlp_block == fused and s.attn_block == fused
        s.fill_cache_random(200, seed=9)
        s.set_state([3], [200])
        s.capture()
        logits = []
        for _ in range(5):
            s.replay()
            torch.cuda.synchronize()
            logits.append(s.logits.clone())
        outs.append((logits, s.ids.clone()))
    for t, (la, lb) in enumerate(zip(outs[0][0], outs[1][0])):
        assert torch.equal(la, lb), f"step {t}: max logit diff {(la - lb).abs().max().item():.3e}"
    assert torch.equal(outs[0][1], outs[1][1])
