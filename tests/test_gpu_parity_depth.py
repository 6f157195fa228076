"""The north star's two parity clauses ASSERTED AT FULL DEPTH:

  (a) "bit-exact token IDs under greedy decode" -- 64 greedy steps (16 for the 80-layer shape) through the captured hipGraph at
      every layer of the BASELINE shapes, ids compared with the oracle's with NO margin filter.  The synthetic model is DECISIVE
      (decoder.build_random_model(decisive=s)): N(0, 0.02) weights give top-2 logit margins of 1e-3 .. 6e-2, below the rounding
      noise ANY two evaluations of a 28-layer bf16 graph have between them (the same oracle in f32 vs f64
      accumulation flips 1 id of 9), so on such a model the clause is a coin toss and not a property of the product.  Here the
      lm_head column of token pi(t) is a multiple of the embedding row of t with the embedding at a few per cent of the final
      hidden state's energy: every layer still moves every logit, but the winner stands well clear -- the test REQUIRES the
      oracle's top-2 margin to be >= 10 x the measured logit error at every step, then requires equal ids, and the ids are
      also the known chain pi^t(last prompt token).
  (b) "logits within 1e-2" -- per-layer teacher-forced drift: the oracle's layer-l input rows and its cache image go into the GPU
      layer l alone (DecodeSession.run_single_layer: the launches of the decode step), the output must agree with the oracle's
      layer output within 1e-2 of its scale -- at EVERY layer.  What is left between the product and the oracle at depth 28 / 80
      is then accumulation of per-layer differences of that size, not a kernel; the end-to-end logit error is asserted
      relative to the logit scale and printed.

The oracle runs layer-major over the whole teacher-forced batch (oracle.model.teacher_forced_trace: one dequantisation per layer,
f32 accumulation as cblas_sgemm carries it, context rows over the fresh K / V, decoder rows over the cache's codec image), fed
the ids the GPU chose.  Reference semantics: batch_mqa_op.cpp:140-179, gemm_op_cpu.cpp:75-126, span_attn_op.cpp:90-169."""
import os

import numpy as np
import pytest
import torch

from oracle import glue, model as omodel

pytestmark = pytest.mark.gpu

CASES = {
    # BASELINE configs[1]-shape headline: Qwen2-7B int4 g128, batch 1, 2048-token history (17-way split attention, GEMV kernels)
    "int4_b1": dict(model="7b", wbits=4, group=128, gptq=False, kv="none", batch=1, prompt=(1984, 1984), steps=64, sigma=3.0,
                    span=128, drift_steps=(0, 63)),
    # configs[2]: batch 32, uint4 KV cache, GPTQ-style integer zeros (small-batch GEMM family, quantised-cache attention)
    "int4_b32_u4kv": dict(model="7b", wbits=4, group=128, gptq=True, kv="u4", batch=32, prompt=(40, 200), steps=64, sigma=3.0,
                          span=32, drift_steps=(0, 63)),
    # configs[3]: one rank's widths of Qwen2-72B at TP = 8, all 80 layers, batch 16
    "cfg3_rank": dict(model="72b_rank", wbits=4, group=128, gptq=False, kv="none", batch=16, prompt=(100, 300), steps=16, sigma=5.0,
                      span=32, drift_steps=(0, 15)),
}


def _cfg(kind):
    from dash_infer_amd import decoder
    if kind == "7b":
        return decoder.QWEN2_7B
    return decoder.ModelConfig("Qwen2-72B rank widths (TP = 8)", hidden=8192, layers=80, n_heads=8, n_kv=1, head_dim=128, inter=3712,
                               vocab=19008)


def _bf16_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).cuda()


@pytest.mark.parametrize("name", list(CASES))
def test_full_depth_greedy_ids_and_per_layer_drift(pkg, name):
    from dash_infer_amd import decoder, ops
    from tests.test_gpu_decoder import _LazyOracleLayers
    c = CASES[name]
    layers_env = os.environ.get("DIHIP_PARITY_DEPTH_LAYERS")   # diagnostics: fewer layers for a quick run
    cfg = _cfg(c["model"])
    nl = int(layers_env) if layers_env else cfg.layers
    seed = 2025
    model = decoder.build_random_model(cfg, decoder.QuantSpec(c["wbits"], c["group"], gptq_like_zeros=c["gptq"]), seed=seed, keep_fp=True,
                                       decisive=c["sigma"], layers=nl)
    perm = decoder.decisive_permutation(cfg.vocab, seed).numpy()
    B, steps, kv_mode = c["batch"], c["steps"], c["kv"]
    rng = np.random.default_rng(B * 101 + steps)
    lens = [int(x) for x in rng.integers(c["prompt"][0], c["prompt"][1] + 1, B)]
    prompts = [[int(t) for t in rng.integers(0, cfg.vocab, n)] for n in lens]
    max_len = max(lens) + steps + 8
    sess = decoder.DecodeSession(model, B, max_len=max_len, span_len=c["span"], kv_mode=kv_mode)

    # ---- the product: context phase, then `steps` greedy steps through the captured hipGraph (what bench.py replays)
    keep = sorted({0, 1, steps // 2, steps - 1})         # decode steps whose full logits rows are kept for the error figure
    lo_gpu = {-1: sess.prefill(prompts).cpu().numpy()}
    gpu_ids = [sess.ids.cpu().numpy().copy()]
    sess.capture(warmup=0)
    gpu_margin = []
    for t in range(steps):
        sess.replay()
        torch.cuda.synchronize()
        top2 = torch.topk(sess.logits, 2, dim=-1).values
        gpu_margin.append(float((top2[:, 0] - top2[:, 1]).min()))
        if t in keep:
            lo_gpu[t] = sess.logits.cpu().numpy().copy()
        gpu_ids.append(sess.ids.cpu().numpy().copy())
    ids = np.stack(gpu_ids)                              # [steps + 1, B]: ids[0] follows the prompt, ids[t + 1] is step t's choice
    chain = np.empty_like(ids)
    cur = np.asarray([p[-1] for p in prompts])
    for t in range(steps + 1):
        cur = perm[cur]
        chain[t] = cur

    # ---- the oracle, layer-major, fed the ids the GPU chose; per layer the GPU layer is teacher-forced on the fly
    f = lambda t: t.float().cpu().numpy()
    olayers = _LazyOracleLayers(model)
    o = omodel.DecoderOracle(olayers, f(model.fp["embed"]), f(model.fp["final_norm"]), f(model.fp["lm_head"]), cfg.n_heads, cfg.n_kv,
                             cfg.head_dim, model.quant.wbits, model.quant.group, eps=cfg.eps, rope_theta=cfg.rope_theta, kv_mode=kv_mode,
                             rounding="x86")
    o.acc = np.float32
    seqs = [list(prompts[b]) + [int(ids[t][b]) for t in range(steps)] for b in range(B)]
    n, g, H = cfg.n_heads, cfg.n_kv, cfg.head_dim
    drift = []                                           # (layer, step, max |err|, max |h_out|, max |err| / max |layer's own contribution|)

    def on_layer(li, h_in, h_out, kvs):
        for s in c["drift_steps"]:
            pos = [lens[b] + s for b in range(B)]        # row of the token fed at decode step s = tokens already cached
            for b in range(B):
                k, v = kvs[b][0][:pos[b]], kvs[b][1][:pos[b]]
                ops.kv_context_copy(sess.kv[li].k_ptrs[b], _bf16_dev(k.reshape(pos[b], g * H)), g * H, pos[b], 0, g, H, c["span"], kv_mode)
                ops.kv_context_copy(sess.kv[li].v_ptrs[b], _bf16_dev(v.reshape(pos[b], g * H)), g * H, pos[b], 0, g, H, c["span"], kv_mode)
            rows_in = np.stack([h_in[b][pos[b]] for b in range(B)])
            rows_out = np.stack([h_out[b][pos[b]] for b in range(B)])
            sess.h.copy_(torch.from_numpy(rows_in).cuda())
            sess.set_state(np.zeros(B, np.int64), pos)
            sess.run_single_layer(li)
            torch.cuda.synchronize()
            got = sess.h.cpu().numpy()
            err = float(np.abs(got - rows_out).max())
            own = float(np.abs(rows_out - rows_in).max())
            drift.append((li, s, err, float(np.abs(rows_out).max()), err / max(own, 1e-30)))

    stats = {"err": {}, "scale": 0.0, "margin": np.full((steps + 1, B), np.inf), "oracle_ids": np.empty((steps + 1, B), np.int64)}

    def on_logits(b, lo):                                # lo [steps + 1, V]: after the prompt, then after each step
        srt = np.partition(lo, -2, axis=-1)[:, -2:]
        stats["margin"][:, b] = srt[:, 1] - srt[:, 0]
        stats["oracle_ids"][:, b] = glue.greedy(lo)
        stats["scale"] = max(stats["scale"], float(np.abs(lo).max()))
        for t, gl in lo_gpu.items():
            e = float(np.abs(gl[b] - lo[t + 1]).max())
            stats["err"][t] = max(stats["err"].get(t, 0.0), e)

    omodel.teacher_forced_trace(o, olayers, seqs, lens, on_layer=on_layer, threads=min(32, os.cpu_count() or 1), want_kv=True,
                                on_logits=on_logits)

    worst_err = max(stats["err"].values())
    min_margin = float(stats["margin"].min())
    mism = int((stats["oracle_ids"] != ids).sum())
    off_chain = int((ids != chain).sum())
    d = np.asarray([(e, sc, rel) for _, _, e, sc, rel in drift])
    worst_layer = drift[int(np.argmax(d[:, 0] / d[:, 1]))]
    report = [
        f"[{name}] {nl} layers, batch {B}, kv {kv_mode}, prompts {min(lens)}..{max(lens)} tokens, {steps} greedy steps (hipGraph), sigma_e {c['sigma']}",
        f"[{name}] greedy ids: {mism}/{ids.size} differ from the oracle's (no margin filter); {off_chain}/{ids.size} off the planted chain",
        f"[{name}] logits: max |err| {worst_err:.3e} at max |logit| {stats['scale']:.2f} = {worst_err / stats['scale']:.2e} relative "
        f"(per kept step: { {t: '%.2e' % e for t, e in sorted(stats['err'].items())} }); oracle top-2 margin min {min_margin:.3f} = "
        f"{min_margin / worst_err:.1f} x the error; product's own min margin {min(gpu_margin):.3f}",
        f"[{name}] per-layer drift (teacher-forced layer, steps {c['drift_steps']}): max over layers of max|err| / max|h_out| = "
        f"{float((d[:, 0] / d[:, 1]).max()):.2e} (layer {worst_layer[0]}, step {worst_layer[1]}: err {worst_layer[2]:.3e}, |h_out| {worst_layer[3]:.2f}); "
        f"relative to the layer's own contribution max {float(d[:, 2].max()):.2e}, median {float(np.median(d[:, 2])):.2e}",
    ]
    per_layer = "\n".join(f"  layer {li:2d} step {s:2d}: max|err| {e:.3e}  max|h_out| {sc:8.3f}  err/|h_out| {e / sc:.2e}  err/|own| {rel:.2e}"
                          for li, s, e, sc, rel in drift)
    print("\n".join(report))
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"parity_depth_{name}.txt"), "w") as fh:
            fh.write("\n".join(report) + "\n" + per_layer + "\n")
    except OSError:
        pass

    # (b) every layer within 1e-2 of its output's scale (teacher-forced): no kernel is off; depth only accumulates
    assert float((d[:, 0] / d[:, 1]).max()) <= 1e-2, report[3]
    # (a) decisive by construction, measured: margin >= 10 x error at EVERY step of EVERY request -- then ids must be equal
    assert min_margin >= 10 * worst_err, report[2]
    assert mism == 0 and off_chain == 0, report[1]
    # end to end, relative to the logit scale (uint4 cache: code steps of range / 15 per element, twice as wide as elsewhere)
    assert worst_err <= (2e-2 if kv_mode == "u4" else 1e-2) * stats["scale"], report[2]
