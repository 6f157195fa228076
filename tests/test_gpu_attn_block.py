"""dihip_decode_attn_block -- RMSNorm + qkv GEMV, Rotary + cache append + paged attention (+ split merge) and the o-projection
+ residual of a batch-1 decode layer in ONE launch -- against the three launches it replaces
(dihip_fused_norm_gemm, dihip_span_attn_decode_fused_sync, dihip_fused_gemm_addto): the hidden row and the cache spans must be
BIT-IDENTICAL (same K split, same partial records, same merge order; csrc/decode_attn_block.hip), eager, back to back (the
launch keeps its own epoch in the sync buffer) and under hipGraph replay.  The chain itself is checked against the oracle in
test_gpu_gemm.py / test_gpu_kv_attn.py / test_gpu_decoder.py.  Reference operators: qwen_v15.py:210-300."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

# Qwen2-7B attention widths (the launch needs >= one column tile per GEMV workgroup: small models keep the chain)
W7B = dict(hidden=3584, layers=2, n_heads=28, n_kv=4, head_dim=128, inter=1024, vocab=2048)


# the weight forms of the launch: (wbits, group) -- int4 g128 (the headline: a group per k-tile), int8 per channel (InstantQuant, BASELINE
# configs[1]), and the other forms of the stand-alone decode GEMV: int8 g64 (a group per k-tile of 64), int8 g128 (two k-tiles per group),
# int4 per channel, int4 g256
FORMS = {"int4g128": (4, 128), "int8perchannel": (8, -1), "int8g64": (8, 64), "int8g128": (8, 128), "int4perchannel": (4, -1), "int4g256": (4, 256)}


def _model(decoder, seed=11, keep_fp=False, w8=False, form=None, **over):
    cfg = decoder.ModelConfig("attn-block", **{**W7B, **over})
    if form is None:
        form = "int8perchannel" if w8 else "int4g128"
    wbits, group = FORMS[form]
    quant = decoder.QuantSpec(wbits, group, gptq_like_zeros=True) if form == "int4g128" else decoder.QuantSpec(wbits, group)
    return decoder.build_random_model(cfg, quant, seed=seed, keep_fp=keep_fp)


def _session(decoder, model, max_len, block):
    old = os.environ.get("DIHIP_DECODER_ATTN_BLOCK")
    os.environ["DIHIP_DECODER_ATTN_BLOCK"] = "1" if block else "0"
    try:
        return decoder.DecodeSession(model, 1, max_len=max_len, span_len=128)
    finally:
        if old is None:
            del os.environ["DIHIP_DECODER_ATTN_BLOCK"]
        else:
            os.environ["DIHIP_DECODER_ATTN_BLOCK"] = old


def _err_word(sess):
    return int(sess.block_sync.view(torch.int32)[1].item())


@pytest.mark.parametrize("form,history", [(f, h) for f in ("int4g128", "int8perchannel") for h in (0, 1, 127, 128, 700, 2047)] +
                         [(f, h) for f in ("int8g64", "int8g128", "int4perchannel", "int4g256") for h in (1, 700, 2047)])
def test_one_launch_equals_the_three_it_replaces(pkg, history, form):
    from dash_infer_amd import decoder, ops
    model = _model(decoder, form=form)
    max_len = 2048 + 64
    a, b = _session(decoder, model, max_len, False), _session(decoder, model, max_len, True)
    assert not a.attn_block and b.attn_block, "the fused launch must serve the Qwen2-7B attention widths on this GPU"
    for s in (a, b):
        s.fill_cache_random(max(history, 1), seed=5)
    b.pool.pool.copy_(a.pool.pool)
    gen = torch.Generator(device="cuda").manual_seed(history + 3)
    h0 = torch.randn(1, model.cfg.hidden, generator=gen, device="cuda", dtype=torch.float32)
    for s in (a, b):
        s.set_state([7], [history])
    for step in range(3):  # three steps back to back: the block's epoch advances, the new rows of step t are history of t + 1
        for s in (a, b):
            s.h.copy_(h0 * (1.0 + 0.25 * step))
            s.run_single_layer(0)
            s.old_lens += 1
            s.new_lens += 1
        torch.cuda.synchronize()
        assert _err_word(b) == 0, "a bounded wait of the fused launch gave up"
        assert torch.equal(a.h, b.h), f"history {history} step {step}: max diff {(a.h - b.h).abs().max().item():.3e}"
        assert torch.isfinite(b.h).all()
    assert torch.equal(a.pool.pool, b.pool.pool), "cache spans differ (DecoderCacheAppend inside the launch)"


@pytest.mark.parametrize("max_len,history", [(3584, 3570), (3008, 2990), (2560, 1000), (1100, 1090), (300, 256)])
def test_other_split_counts_are_bit_identical_too(pkg, max_len, history):
    """The split count follows max_len (one split per 128 tokens up to what the CUs hold): 28 / 24 / 20 / 9 / 3 splits -- the block's quad-lane
    merge polls 7 / 6 / 5 / 3 / 1 records per lane; the chain's in-launch merge loads one batch of <= 24 records per lane, and at 28 splits takes
    the maximum pass first and then batches of 16 -- all bit-identical: every split merge sums in merge_order4 (csrc/span_attn_common.hpp)."""
    from dash_infer_amd import decoder
    model = _model(decoder, seed=13)
    a, b = _session(decoder, model, max_len, False), _session(decoder, model, max_len, True)
    assert b.attn_block, f"max_len {max_len}: the block must serve this split count"
    for s in (a, b):
        s.fill_cache_random(max(history, 1), seed=4)
    b.pool.pool.copy_(a.pool.pool)
    gen = torch.Generator(device="cuda").manual_seed(max_len)
    h0 = torch.randn(1, model.cfg.hidden, generator=gen, device="cuda", dtype=torch.float32)
    for s in (a, b):
        s.set_state([11], [history])
    for step in range(3):
        for s in (a, b):
            s.h.copy_(h0 * (1.0 + 0.5 * step))
            s.run_single_layer(step % 2)
            s.old_lens += 1
            s.new_lens += 1
        torch.cuda.synchronize()
        assert _err_word(b) == 0
        assert torch.equal(a.h, b.h), f"max_len {max_len} step {step}: max diff {(a.h - b.h).abs().max().item():.3e}"
    assert torch.equal(a.pool.pool, b.pool.pool)


@pytest.mark.parametrize("w8", [False, True], ids=["int4g128", "int8perchannel"])
def test_decode_steps_through_a_replayed_graph_are_bit_identical(pkg, w8):
    """whole decode steps (2 layers, final norm, lm_head, greedy) through a captured hipGraph, replayed: logits of every step equal"""
    from dash_infer_amd import decoder
    model = _model(decoder, seed=23, w8=w8)
    max_len = 512
    outs = []
    for block in (False, True):
        s = _session(decoder, model, max_len, block)
        assert s.attn_block == block
        s.fill_cache_random(200, seed=9)
        s.set_state([3], [200])
        s.capture()
        logits = []
        for _ in range(6):
            s.replay()
            torch.cuda.synchronize()
            logits.append(s.logits.clone())
        if block:
            assert _err_word(s) == 0
        outs.append((logits, s.ids.clone(), s.pool.pool.clone()))
    for t, (la, lb) in enumerate(zip(outs[0][0], outs[1][0])):
        assert torch.equal(la, lb), f"step {t}: max logit diff {(la - lb).abs().max().item():.3e}"
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_unsupported_configurations_keep_the_chain(pkg):
    from dash_infer_amd import decoder
    small = decoder.ModelConfig("small", hidden=512, layers=1, n_heads=4, n_kv=2, head_dim=128, inter=1024, vocab=1024)
    m = decoder.build_random_model(small, decoder.QuantSpec(4, 128), seed=3)
    assert not _session(decoder, m, 64, True).attn_block           # fewer column tiles than GEMV workgroups
    m4 = _model(decoder)
    assert not decoder.DecodeSession(m4, 2, max_len=256, span_len=128).attn_block                     # batch 2
    assert not decoder.DecodeSession(m4, 1, max_len=256, span_len=128, kv_mode="u4").attn_block       # quantised cache


def test_cpp_operator_list_runs_the_block_and_matches_decode_session(pkg):
    """The C++ operator layer (host/fused_ops_hip.cpp): DihipNormGemm, DihipRopeSpanAttn and DihipGemmAddTo stay three operators; for
    one request the o-projection issues the single launch and the other two skip theirs.  Logits bit-identical to DecodeSession with
    the block on AND to the three-launch chain (DIHIP_DECODER_ATTN_BLOCK=0 is read once per process on the C++ side: the chain is
    covered through the Python session here), context phase + graph-replayed steps; then a second request joins (batch 2: chain)."""
    from dash_infer_amd import decoder
    from tests.test_gpu_host_runner import Host
    model = _model(decoder, seed=31, keep_fp=True)   # (the operator layer re-lays-out the unpacked weights itself)
    cfg = model.cfg
    span, max_len, steps = 128, 384, 5
    prompt = [int(t) for t in torch.randint(0, cfg.vocab, (150,), generator=torch.Generator().manual_seed(1)).tolist()]
    want = {}
    for block in (False, True):
        s = _session(decoder, model, max_len, block)
        assert s.attn_block == block
        lo0 = s.prefill([prompt]).clone()
        out = []
        for _ in range(steps):
            s.step()
            torch.cuda.synchronize()
            out.append((s.logits.clone(), s.ids.cpu().tolist()))
        want[block] = (lo0, out)
    for t in range(steps):
        assert torch.equal(want[False][1][t][0], want[True][1][t][0]), f"python runner, step {t}: block != chain"
    h = Host(model, 2, max_len, span, "none")
    assert h.report["fused"], h.report["why"]
    k, v = h.spans()
    h.start(prompt, k, v)
    assert torch.equal(h.logits()[0], want[True][0][0])
    for t in range(steps):
        ids = h.steps(1, graph=True)
        assert ids == want[True][1][t][1], f"step {t}: ids differ"
        assert torch.equal(h.logits(), want[True][1][t][0]), f"step {t}: logits are not bit-identical to DecodeSession"
    # a second request joins: batch 2 runs the three launches again (Reshape re-decides), then leaves
    k2, v2 = h.spans()
    h.start(prompt[:40], k2, v2)
    h.steps(2, graph=True)
    h.close()


def test_a_rank_that_does_not_carry_the_residual(pkg):
    """Under tensor parallelism only rank 0 adds the residual (gemm_op.cpp:133-137): the other ranks call the launch with h_res = NULL and must
    get exactly h_res = 0 (the all-reduce behind it sums the ranks' rows).  Same launch, same cache append."""
    from dash_infer_amd import decoder, ops
    model = _model(decoder, seed=41)
    cfg = model.cfg
    outs = []
    for null_res in (True, False):
        s = _session(decoder, model, 512, True)
        assert s.attn_block
        s.fill_cache_random(300, seed=2)
        s.set_state([5], [300])
        gen = torch.Generator(device="cuda").manual_seed(8)
        h_in = torch.randn(1, cfg.hidden, generator=gen, device="cuda", dtype=torch.float32)
        lw = model.layers[0]
        out = torch.full_like(h_in, float("nan"))
        res = None if null_res else torch.zeros_like(h_in)
        ops.decode_attn_block(h_in, res, lw.ln1, cfg.eps, lw.qkv, lw.qkv_bias, lw.o, s.kv[0], s.old_lens, s.rope_tab, s.n_loc, s.g_loc, s.H, s.max_len,
                              s.scale, s.attn_ws, s.block_sync, out=out)
        torch.cuda.synchronize()
        assert _err_word(s) == 0 and torch.isfinite(out).all()
        outs.append((out.clone(), s.pool.pool.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_a_timed_out_handoff_is_reported_and_the_chain_takes_over(pkg, monkeypatch):
    """a bounded wait that gives up must never become silently wrong tokens.  DIHIP_ATTN_BLOCK_FAULT=1 makes one
    GEMV workgroup withhold its qkv rows (and shortens the spin limit): every consumer times out, the launch sets the error word and
    returns garbage.  DecodeSession.check_handoffs() -- the caller's synchronisation point -- raises, restores the buffer and leaves the
    session on the three-launch chain: the SAME steps decoded again equal the chain's, bit for bit."""
    from dash_infer_amd import decoder
    model = _model(decoder, seed=41)
    max_len = 512
    ref = _session(decoder, model, max_len, False)
    ref.fill_cache_random(180, seed=2)
    ref.set_state([5], [180])
    want = []
    for _ in range(3):
        ref.step()
        torch.cuda.synchronize()
        want.append((ref.logits.clone(), ref.ids.clone()))
    s = _session(decoder, model, max_len, True)
    assert s.attn_block
    s.fill_cache_random(180, seed=2)
    s.set_state([5], [180])
    s.step()                       # a good step first: the epoch and the record buffers are in their steady state
    torch.cuda.synchronize()
    s.check_handoffs()
    assert torch.equal(s.logits, want[0][0])
    monkeypatch.setenv("DIHIP_ATTN_BLOCK_FAULT", "1")
    s.step()
    monkeypatch.delenv("DIHIP_ATTN_BLOCK_FAULT")
    with pytest.raises(RuntimeError, match="hand-off wait gave up"):
        s.check_handoffs()
    assert not s.attn_block and _err_word(s) == 0
    # the caller re-runs from its last good state (ids / lengths of step 1) on the chain
    s.set_state(want[0][1].tolist(), [181])
    for t in (1, 2):
        s.step()
        torch.cuda.synchronize()
        s.check_handoffs()
        assert torch.equal(s.logits, want[t][0]) and torch.equal(s.ids, want[t][1]), f"step {t} after the recovery"


def test_cpp_runner_reads_the_error_word_at_sync_and_recovers(pkg, monkeypatch):
    """The same through the C++ operator layer: HipModelRunner::Sync reads the error word beside the id readback; a timed-out launch
    fails THAT synchronise (AsStatus RUNTIME_ERROR, message names the hand-off), the requests are rolled back to the last good
    synchronise, the attention block is switched off for the context and the next decode_steps decodes the same tokens on the chain."""
    from dash_infer_amd import decoder, hostapi
    from tests.test_gpu_host_runner import Host
    model = _model(decoder, seed=43, keep_fp=True)
    cfg = model.cfg
    span, max_len = 128, 384
    prompt = [int(t) for t in torch.randint(0, cfg.vocab, (90,), generator=torch.Generator().manual_seed(2)).tolist()]
    ref = _session(decoder, model, max_len, False)
    ref.prefill([prompt])
    want = []
    for _ in range(4):
        ref.step()
        torch.cuda.synchronize()
        want.append(ref.ids.cpu().tolist())
    h = Host(model, 1, max_len, span, "none")
    assert h.report["fused"], h.report["why"]
    k, v = h.spans()
    h.start(prompt, k, v)
    assert h.steps(1, graph=False) == want[0]
    monkeypatch.setenv("DIHIP_ATTN_BLOCK_FAULT", "1")
    with pytest.raises(hostapi.HostError, match="hand-off wait gave up"):
        h.steps(2, graph=False)
    monkeypatch.delenv("DIHIP_ATTN_BLOCK_FAULT")
    # rolled back to the state after step 0; the chain decodes steps 1 .. 3 (eager, then replayed from a fresh capture)
    assert h.steps(1, graph=False) == want[1]
    assert h.steps(1, graph=True) == want[2]
    assert h.steps(1, graph=True) == want[3]
    h.close()


def test_block_under_load_is_bit_identical_and_never_times_out(pkg):
    """The launch's hand-offs (tagged granules, polled split records with their two parity buffers) under UNEVEN load with warm caches
    (MI355X guide: the conditions that expose a stale or torn hand-off): 200 launches back to back over four alternating hidden rows -- so
    that a granule or record left over from the previous launch would carry other data -- while a second stream saturates HBM and a
    third issues a stream of tiny kernels.  Every launch equals the three-launch chain's bits for its row; the error word stays zero."""
    from dash_infer_amd import decoder
    from tests.test_gpu_gemv_stress import Load, REPS
    model = _model(decoder, seed=51)
    max_len = 2048 + 64
    a, b = _session(decoder, model, max_len, False), _session(decoder, model, max_len, True)
    assert b.attn_block
    for s in (a, b):
        s.fill_cache_random(1777, seed=6)
    b.pool.pool.copy_(a.pool.pool)
    for s in (a, b):
        s.set_state([9], [1777])
    gen = torch.Generator(device="cuda").manual_seed(77)
    rows = [torch.randn(1, model.cfg.hidden, generator=gen, device="cuda", dtype=torch.float32) * (0.5 + 0.5 * i) for i in range(4)]
    want = []
    for h0 in rows:                      # the chain on the idle GPU (the appended K / V row of position 1777 is rewritten each time)
        a.h.copy_(h0)
        a.run_single_layer(0)
        torch.cuda.synchronize()
        want.append(a.h.clone())
    keep = torch.empty((REPS,) + tuple(want[0].shape), dtype=torch.float32, device="cuda")
    load = Load()
    load.enqueue()
    for r in range(REPS):
        b.h.copy_(rows[(r // 2) % 4], non_blocking=True)
        b.run_single_layer(r % 2)        # both layers' weights, one sync buffer: the epoch and the record parity advance per launch
        if r % 2 == 0:
            keep[r].copy_(b.h, non_blocking=True)
    torch.cuda.synchronize()
    load.wait()
    assert _err_word(b) == 0, "a bounded wait gave up under load"
    bad = [r for r in range(0, REPS, 2) if not torch.equal(keep[r], want[(r // 2) % 4])]
    assert not bad, f"{len(bad)} launches differ from the chain (first: repetition {bad[0]})"
    b.check_handoffs()


def test_the_eight_wave_attention_form_in_a_process_of_its_own():
    """DIHIP_ATTN_WIDE=1 (read once per process): 8-wave attention workgroups at batch 1 -- half the split count -- in the stand-alone kernel
    (span_attn_ft_mfma_w8_kernel) AND in the block (decode_attn_block_kernel<8>): the bit-identity tests of this file and the oracle
    comparisons of the 16-bit cache's decode attention, re-run in a child process with the switch on."""
    import subprocess
    import sys
    if os.environ.get("DIHIP_ATTN_WIDE") == "1":
        pytest.skip("already the child")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, DIHIP_ATTN_WIDE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_gpu_attn_block.py"), os.path.join(here, "test_gpu_kv_attn.py"),
                        "-k", "one_launch_equals or other_split_counts or replayed_graph or span_attention_unquantised or fused_rope_append or long_context"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here))
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


def test_launch_plans_follow_the_requests_length_not_the_engines_maximum(pkg):
    """host/model_runner.cpp DecodeSteps + HIPContext::PlanLength: the decode step's launch plans (attention split count / width, the fused block's
    grid) are made for the running requests' length rounded up to a 512-token bucket, not for the engine's maximum length.  One request of
    ~1000 tokens decodes across the 1024 bucket boundary in an engine of 4096 tokens and in one of 1536: the plans are the same (1024, then 1536),
    so logits and ids are BIT-IDENTICAL at every step -- with plans made for the maximum length they could not be (32 against 12 splits) and the
    4096-token engine could not run the attention block at all (its 32 x 4 attention workgroups leave too few for the GEMVs); eager steps and
    the re-captured graph agree as well."""
    from dash_infer_amd import decoder
    from tests.test_gpu_host_runner import Host
    model = _model(decoder, seed=37, keep_fp=True)
    cfg = model.cfg
    span, steps = 128, 8
    prompt = [int(t) for t in torch.randint(0, cfg.vocab, (1019,), generator=torch.Generator().manual_seed(2)).tolist()]
    from dash_infer_amd import ops
    sup = lambda n: ops.decode_attn_block_supported(model.layers[0].qkv, cfg.hidden, cfg.n_heads, cfg.n_kv, cfg.head_dim, n, "none", torch.bfloat16, 1)
    assert not sup(4096) and sup(1024) and sup(1536)
    runs = {}
    for tag, max_len, graph in (("4096 graph", 4096, True), ("1536 graph", 1536, True), ("4096 eager", 4096, False)):
        h = Host(model, 1, max_len, span, "none")
        assert h.report["fused"], h.report["why"]
        k, v = h.spans()
        first = h.start(prompt, k, v)
        out = []
        for _ in range(steps):    # lengths 1020 .. 1027: the bucket moves from 1024 to 1536 after the fifth step
            ids = h.steps(1, graph=graph)
            out.append((h.logits().float().clone(), ids))
        runs[tag] = (first, out)
        h.close()
    for other in ("1536 graph", "4096 eager"):
        assert runs["4096 graph"][0] == runs[other][0]
        for t, ((la, ia), (lb, ib)) in enumerate(zip(runs["4096 graph"][1], runs[other][1])):
            assert ia == ib, f"{other}, step {t}: ids differ"
            assert torch.equal(la, lb), f"{other}, step {t}: max diff {(la - lb).abs().max().item():.3e}"
    # ... and across 3584 tokens, where the bucket's plan (4096: 32 splits) no longer fits the block: the step goes on through the three launches
    # (Reshape decides anew), captured again -- graph and eager agree bit for bit on both sides of the boundary
    assert sup(3584) and not sup(4096)
    prompt = [int(t) for t in torch.randint(0, cfg.vocab, (3580,), generator=torch.Generator().manual_seed(3)).tolist()]
    runs = {}
    for graph in (True, False):   # (graph first: its first captured step meets a sync buffer whose address an earlier runner used with another plan)
        h = Host(model, 1, 4096, span, "none")
        k, v = h.spans()
        first = h.start(prompt, k, v)
        out = []
        for _ in range(7):        # lengths 3581 .. 3587
            ids = h.steps(1, graph=graph)
            out.append((h.logits().float().clone(), ids))
        runs[graph] = (first, out)
        h.close()
    assert runs[True][0] == runs[False][0]
    for t, ((la, ia), (lb, ib)) in enumerate(zip(runs[True][1], runs[False][1])):
        assert ia == ib and torch.equal(la, lb), f"across 3584, step {t}"


def test_a_changed_split_plan_is_cleared_outside_a_capture_or_refused_inside(pkg):
    """The polled records of a sync buffer are laid out per split plan; a launch with another plan clears the record region first -- eagerly.
    Inside a stream capture it must not (a memset captured into the graph ran at every replay and corrupted replayed steps: round 6): the
    launch fails and names dihip_decode_attn_block_prepare(), which the capturing caller runs outside the capture (the C++ operator layer
    does at Reshape); after it the captured launch is the eager one, bit for bit."""
    from dash_infer_amd import decoder, ops
    from dash_infer_amd.capi import lib
    model = _model(decoder, seed=41)
    a, b = _session(decoder, model, 1100, False), _session(decoder, model, 1100, True)
    for s in (a, b):
        s.fill_cache_random(200, seed=6)
        s.set_state([5], [200])
    b.pool.pool.copy_(a.pool.pool)
    h0 = torch.randn(1, model.cfg.hidden, generator=torch.Generator(device="cuda").manual_seed(9), device="cuda", dtype=torch.float32)
    b.max_len = 300          # a first launch with the 3-split plan of a 300-token engine on this sync buffer ...
    b.h.copy_(h0)
    b.run_single_layer(0)
    torch.cuda.synchronize()
    b.max_len = 1100         # ... then the 9-split plan, under capture: refused
    b.set_state([5], [200])
    b.pool.pool.copy_(a.pool.pool)
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(Exception) as e:
        with torch.cuda.stream(st):
            b.h.copy_(h0)
            with torch.cuda.graph(g, stream=st):
                b.run_single_layer(0)
    assert "prepare" in str(e.value)
    torch.cuda.synchronize()
    cfg = model.cfg
    ops.check(lib().dihip_decode_attn_block_prepare(ops.cur_stream(), ops.ptr(b.block_sync), b.block_sync.numel(), cfg.n_heads, cfg.n_kv, cfg.head_dim, 1100),
              "prepare")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        b.h.copy_(h0)
        with torch.cuda.graph(g, stream=st):
            b.run_single_layer(0)
        b.h.copy_(h0)
        g.replay()
    torch.cuda.synchronize()
    a.h.copy_(h0)
    a.run_single_layer(0)
    torch.cuda.synchronize()
    assert _err_word(b) == 0
    assert torch.equal(a.h, b.h), f"max diff {(a.h - b.h).abs().max().item():.3e}"
