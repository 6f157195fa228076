"""GPU parity of the C++ operator layer (AsOperator subclasses on DeviceType::HIP), driven like
AsModel drives ops: CallInit -> CallReshape -> [CallAlloc] -> CallForward, tensors bound by name,
shared "workspace", RuntimeContext with per-request span vectors.  Reference: the oracle."""
import numpy as np
import pytest
import torch

from oracle import attention, gemm_ref, glue, kv_codec, quant
from oracle.numerics import bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(pkg):
    from dash_infer_amd import hostapi, ops
    assert torch.cuda.is_available()
    return hostapi, ops


def dev(a, dt=torch.bfloat16):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dt).cuda()


def view_of(ptr, shape, dtype):
    """Copy of model-owned device memory as a torch tensor (torch has no from-pointer constructor)."""
    import ctypes as C
    n = int(np.prod(shape))
    t = torch.empty(n, dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), n * t.element_size(), 3) == 0
    return t.view(*shape)


@pytest.mark.parametrize("wbits,G,act", [(4, 128, None), (8, -1, "silu"), (4, 64, "gelu_erf")])
def test_gemm_lowp_operator(env, wbits, G, act):
    hostapi, ops = env
    rng = np.random.default_rng(wbits + abs(G))
    K, N = 512, 384
    W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
    q, s, z = (quant.iq_quantize_a16w4(W, G, "bf16") if wbits == 4 else quant.iq_quantize_a16w8(W, G, "bf16"))
    bias = bf16_round(rng.normal(0, 0.2, N).astype(np.float32))
    m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16)
    qd = torch.from_numpy(q).cuda()
    m.set_weight("w", qd, "u8" if wbits == 4 else "i8")
    m.set_weight("w.scales", dev(s), "bf16")
    m.set_weight("w.zeros", dev(z), "bf16")
    m.set_weight("w.bias", dev(bias), "bf16")
    actcode = {None: 0, "silu": 5, "gelu_erf": 2}[act]
    attrs = f"alpha=f:1.0;activation=i:{actcode}" + (f";GroupSize=i:{G}" if G > 0 else "")
    for M in (1, 5, 40):  # batch changes -> Reshape again, same op instance
        x = bf16_round(rng.normal(0, 1, (M, 1, K)).astype(np.float32))
        xd = dev(x)
        m.set_tensor("x", xd, "bf16")
        if M == 1:
            op = m.create_op("GemmA16W4" if wbits == 4 else "GemmA16W8", "decoder.layer.0.ffn.gate", ["x"], ["y"],
                             ["w", "w.scales", "w.zeros", "w.bias"], attrs)
        m.reshape(op)
        m.alloc(op)
        m.forward(op)
        dt, shape, ptr = m.get_tensor("y")
        assert dt == hostapi.DT["bf16"] and shape == [M, 1, N]
        y = view_of(ptr, shape, torch.bfloat16).float().cpu().numpy().reshape(M, N)
        ref = gemm_ref.gemm_a16wx(x.reshape(M, K), q, s, z, G, wbits, bias=bias, act=act, ft="bf16")
        np.testing.assert_allclose(y, ref, rtol=2 ** -7, atol=2 ** -7 * 0.05 * np.abs(ref).max())
    m.close()


def test_gemm_operator_rejects_bad_attributes(env):
    hostapi, ops = env
    m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16)
    z8 = torch.zeros(64, 32, dtype=torch.int8, device="cuda")
    m.set_weight("w", z8, "i8")
    m.set_weight("s", torch.zeros(1, 32, dtype=torch.bfloat16, device="cuda"), "bf16")
    m.set_weight("z", torch.zeros(1, 32, dtype=torch.bfloat16, device="cuda"), "bf16")
    m.set_tensor("x", torch.zeros(1, 1, 64, dtype=torch.bfloat16, device="cuda"), "bf16")
    for bad in ("transB=b:1", "is_pooler=b:1", "GroupSize=i:48"):
        with pytest.raises(hostapi.HostError) as e:
            m.create_op("GemmA16W8", "g", ["x"], ["y"], ["w", "s", "z"], bad)
        assert e.value.code == 2
    m.close()


@pytest.mark.parametrize("mode", ["none", "u4"])
def test_span_attention_operator_prefill_then_decode(env, mode):
    """DecOptMQA: a 70-token prefill (runContext: MFMA prefill attention + ContextSpanCopy), then two
    decode steps of a batch of two requests (runDecoder: cache append + paged attention)."""
    hostapi, ops = env
    rng = np.random.default_rng(9)
    n, g, H, S, L = 8, 2, 128, 32, 70
    max_len = 128
    spr = max_len // S
    cache_mode = {"none": 0, "i8": 1, "u4": 2}[mode]
    pool = ops.SpanPool(2 * 2 * spr + 1, g, S, H, mode, torch.bfloat16)
    m = hostapi.Model(ops.cur_stream(), n, g, H, S, cache_mode, max_batch=2, max_len=max_len)
    alpha = 1.0 / np.sqrt(H)
    reqs = []
    op = None
    for r in range(2):
        kp = [pool.alloc()[0] for _ in range(spr)]
        vp_ = [pool.alloc()[0] for _ in range(spr)]
        qkv = bf16_round(rng.normal(0, 1, (L, (n + 2 * g) * H)).astype(np.float32))
        xd = dev(qkv.reshape(1, L, -1))
        m.set_tensor("qkv", xd, "bf16")
        if op is None:
            op = m.create_op("DecOptMQA", "decoder.layer.0.attention", ["qkv"], ["attn_out"])
        m.set_runtime(True, [0], [[kp]], [[vp_]])
        m.reshape(op)
        m.alloc(op)
        m.forward(op)
        _, shape, ptr = m.get_tensor("attn_out")
        out = view_of(ptr, shape, torch.bfloat16).float().cpu().numpy().reshape(L, n, H)
        q = qkv[:, : n * H].reshape(L, n, H)
        k = qkv[:, n * H:(n + g) * H].reshape(L, g, H)
        v = qkv[:, (n + g) * H:].reshape(L, g, H)
        np.testing.assert_allclose(out, attention.prefill_attention(q, k, v, alpha), rtol=1e-2, atol=5e-3)  # P is bf16 on the matrix core
        kc, vc = kv_codec.SpanCache(g, S, H, mode, "bf16"), kv_codec.SpanCache(g, S, H, mode, "bf16")
        for t in range(L):
            kc.write(t, k[t])
            vc.write(t, v[t])
        reqs.append((kp, vp_, kc, vc))
    # the spans now hold exactly what the codec oracle holds
    torch.cuda.synchronize()
    for (kp, vp_, kc, vc) in reqs:
        for i, sp in enumerate(kc.spans):
            idx = (kp[i] - pool.pool.data_ptr()) // pool.aligned
            assert np.array_equal(pool.span_view(idx).cpu().numpy(), sp)
    # two decode steps of the batch of two
    for step in range(2):
        qkv = bf16_round(rng.normal(0, 1, (2, (n + 2 * g) * H)).astype(np.float32))
        m.set_tensor("qkv", dev(qkv.reshape(2, 1, -1)), "bf16")
        m.set_runtime(False, [L + step, L + step], [[r[0]] for r in reqs], [[r[1]] for r in reqs])
        m.reshape(op)
        m.alloc(op)
        m.forward(op)
        _, shape, ptr = m.get_tensor("attn_out")
        out = view_of(ptr, shape, torch.bfloat16).float().cpu().numpy().reshape(2, n, H)
        for b, (kp, vp_, kc, vc) in enumerate(reqs):
            kc.write(L + step, qkv[b, n * H:(n + g) * H].reshape(g, H))
            vc.write(L + step, qkv[b, (n + g) * H:].reshape(g, H))
            ref = attention.decode_attention(qkv[b, : n * H].reshape(n, H), kc.read_all(L + step + 1), vc.read_all(L + step + 1), alpha)
            np.testing.assert_allclose(out[b], ref, rtol=1e-2, atol=2.5e-3)
    m.close()


def test_allreduce_operator_single_rank(env):
    hostapi, ops = env
    m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16)
    x = torch.randn(3, 1, 64, device="cuda").to(torch.bfloat16)
    m.set_tensor("x", x, "bf16")
    op = m.create_op("AllReduce", "decoder.layer.0.allreduce", ["x"], ["y"])
    m.reshape(op)
    m.alloc(op)
    m.forward(op)
    _, shape, ptr = m.get_tensor("y")
    assert torch.equal(view_of(ptr, shape, torch.bfloat16), x)
    m.close()


@pytest.mark.parametrize("mode", ["none", "i8"])
def test_span_attention_operator_prefill_over_cached_prefix(env, mode):
    """Prefix-cache hit: request A prefills 64 tokens; request B shares A's first two
    spans (prefix_len = 64) and prefills 40 more.  B's output must equal the last 40 rows of attention over
    [dequantised cached prefix | new rows], and B's new spans must hold the codec image of its new rows."""
    hostapi, ops = env
    rng = np.random.default_rng(21)
    n, g, H, S = 8, 2, 128, 32
    P, Lnew, max_len = 64, 40, 160
    spr = max_len // S
    cache_mode = {"none": 0, "i8": 1}[mode]
    pool = ops.SpanPool(4 * spr + 1, g, S, H, mode, torch.bfloat16)
    m = hostapi.Model(ops.cur_stream(), n, g, H, S, cache_mode, max_batch=1, max_len=max_len)
    alpha = 1.0 / np.sqrt(H)
    kpA = [pool.alloc()[0] for _ in range(spr)]
    vpA = [pool.alloc()[0] for _ in range(spr)]
    qkvA = bf16_round(rng.normal(0, 1, (P, (n + 2 * g) * H)).astype(np.float32))
    m.set_tensor("qkv", dev(qkvA.reshape(1, P, -1)), "bf16")
    op = m.create_op("DecOptMQA", "decoder.layer.0.attention", ["qkv"], ["attn_out"])
    m.set_runtime(True, [0], [[kpA]], [[vpA]])
    m.reshape(op)
    m.alloc(op)
    m.forward(op)
    # request B: same first two spans, fresh ones behind
    kpB = kpA[: P // S] + [pool.alloc()[0] for _ in range(spr - P // S)]
    vpB = vpA[: P // S] + [pool.alloc()[0] for _ in range(spr - P // S)]
    qkvB = bf16_round(rng.normal(0, 1, (Lnew, (n + 2 * g) * H)).astype(np.float32))
    m.set_tensor("qkv", dev(qkvB.reshape(1, Lnew, -1)), "bf16")
    m.set_runtime(True, [P], [[kpB]], [[vpB]])  # the request's cache already holds the shared prefix (gen_ctx->step == prefix_len)
    m.set_prefix_len(0, P)
    m.reshape(op)
    m.alloc(op)
    m.forward(op)
    _, shape, ptr = m.get_tensor("attn_out")
    out = view_of(ptr, shape, torch.bfloat16).float().cpu().numpy().reshape(Lnew, n, H)
    # oracle: cached prefix as the codec decodes it, then the new rows
    kc, vc = kv_codec.SpanCache(g, S, H, mode, "bf16"), kv_codec.SpanCache(g, S, H, mode, "bf16")
    kA = qkvA[:, n * H:(n + g) * H].reshape(P, g, H)
    vA = qkvA[:, (n + g) * H:].reshape(P, g, H)
    for t in range(P):
        kc.write(t, kA[t])
        vc.write(t, vA[t])
    kB = qkvB[:, n * H:(n + g) * H].reshape(Lnew, g, H)
    vB = qkvB[:, (n + g) * H:].reshape(Lnew, g, H)
    kfull = np.concatenate([bf16_round(kc.read_all(P)), kB])
    vfull = np.concatenate([bf16_round(vc.read_all(P)), vB])
    ref = attention.prefill_attention(qkvB[:, : n * H].reshape(Lnew, n, H), kfull, vfull, alpha)
    np.testing.assert_allclose(out, ref, rtol=1e-2, atol=5e-3)
    for t in range(Lnew):
        kc.write(P + t, kB[t])
        vc.write(P + t, vB[t])
    torch.cuda.synchronize()
    for i in range(P // S, (P + Lnew + S - 1) // S):
        idx = (kpB[i] - pool.pool.data_ptr()) // pool.aligned
        got = pool.span_view(idx).cpu().numpy()
        exp = kc.spans[i]
        if i == (P + Lnew) // S and (P + Lnew) % S:  # last span partially filled: compare written positions only
            continue
        assert np.array_equal(got, exp), f"K span {i}"
    m.close()


def test_moe_a16w8_operator(env):
    """Op type MOEA16W8 through the AsOperator interface (host/moe_op_hip.cpp): stacked int8 experts with [gate | up]
    columns, routing from the second input, shared workspace; against oracle/moe.py; batch change -> Reshape again."""
    from oracle import moe
    hostapi, ops = env
    rng = np.random.default_rng(8)
    E, k, hidden, proj, G = 6, 2, 256, 384, -1
    def experts(K, N):
        qs, ss, zs = [], [], []
        for _ in range(E):
            W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
            q, s, z = quant.iq_quantize_a16w8(W, G, "bf16")
            qs.append(q); ss.append(s); zs.append(z)
        return np.stack(qs), np.stack(ss), np.stack(zs)
    gq, gs, gz = experts(hidden, proj)
    uq, us, uz = experts(hidden, proj)
    dq, ds, dz = experts(proj, hidden)
    m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16)
    m.set_weight("gu", torch.from_numpy(np.concatenate([gq, uq], axis=2)).cuda(), "i8")       # [E, hidden, 2*proj]
    m.set_weight("gu.scales", dev(np.concatenate([gs, us], axis=2)), "bf16")
    m.set_weight("gu.zeros", dev(np.concatenate([gz, uz], axis=2)), "bf16")
    m.set_weight("dn", torch.from_numpy(dq).cuda(), "i8")
    m.set_weight("dn.scales", dev(ds), "bf16")
    m.set_weight("dn.zeros", dev(dz), "bf16")
    op = None
    for T in (1, 5):
        x = bf16_round(rng.normal(0, 1, (T, 1, hidden)).astype(np.float32))
        logits = bf16_round(rng.normal(0, 1.5, (T, 1, E)).astype(np.float32))
        m.set_tensor("x", dev(x), "bf16")
        m.set_tensor("router", dev(logits), "bf16")
        if op is None:
            op = m.create_op("MOEA16W8", "decoder.layer.0.mlp.moe", ["x", "router"], ["y"],
                             ["gu", "gu.scales", "gu.zeros", "dn", "dn.scales", "dn.zeros"], f"num_experts=i:{E};num_experts_per_tok=i:{k}")
        m.reshape(op)
        m.alloc(op)
        m.forward(op)
        dt, shape, ptr = m.get_tensor("y")
        assert dt == hostapi.DT["bf16"] and shape == [T, 1, hidden]
        y = view_of(ptr, shape, torch.bfloat16).float().cpu().numpy().reshape(T, hidden)
        s_ref, e_ref = moe.route(logits.reshape(T, E), k)
        ref = moe.experts_ffn(x.reshape(T, hidden), e_ref, s_ref, list(zip(gq, gs, gz)), list(zip(uq, us, uz)), list(zip(dq, ds, dz)), G, 8)
        np.testing.assert_allclose(y, ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())
    with pytest.raises(hostapi.HostError):   # attribute checks of MoeOp::Init (moe_op.cpp:65-80)
        m.create_op("MOEA16W8", "bad", ["x", "router"], ["y2"], ["gu", "gu.scales", "gu.zeros", "dn", "dn.scales", "dn.zeros"],
                    f"num_experts=i:{E}")
    m.close()


def test_moe_expert_parallel_ranks_sum_to_the_whole_and_calc_expert(env):
    """MOEA16W8 with attribute use_ep (moe_op.cpp:103-117): rank r holds the stack of experts [r * E / nranks, ...) and computes
    only their terms; the ranks' outputs sum (the AllReduce after the operator) to the single-rank result.  Then op type
    CalcExpert (calc_expert_op.cpp:14-68): rows scaled by a per-token weight, against numpy."""
    from oracle import moe
    hostapi, ops = env
    rng = np.random.default_rng(21)
    E, k, hidden, proj, G, T, nranks = 8, 3, 256, 256, -1, 4, 2
    def experts(K, N):
        qs, ss, zs = [], [], []
        for _ in range(E):
            q, s_, z = quant.iq_quantize_a16w8(bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32)), G, "bf16")
            qs.append(q); ss.append(s_); zs.append(z)
        return np.stack(qs), np.stack(ss), np.stack(zs)
    gq, gs, gz = experts(hidden, proj)
    uq, us, uz = experts(hidden, proj)
    dq, ds, dz = experts(proj, hidden)
    x = bf16_round(rng.normal(0, 1, (T, 1, hidden)).astype(np.float32))
    logits = bf16_round(rng.normal(0, 1.5, (T, 1, E)).astype(np.float32))
    total = np.zeros((T, hidden), np.float32)
    for rank in range(nranks):
        lo, hi = rank * E // nranks, (rank + 1) * E // nranks
        m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16, rank=rank, nranks=nranks)
        m.set_weight("gu", torch.from_numpy(np.concatenate([gq[lo:hi], uq[lo:hi]], axis=2)).cuda(), "i8")
        m.set_weight("gu.scales", dev(np.concatenate([gs[lo:hi], us[lo:hi]], axis=2)), "bf16")
        m.set_weight("gu.zeros", dev(np.concatenate([gz[lo:hi], uz[lo:hi]], axis=2)), "bf16")
        m.set_weight("dn", torch.from_numpy(dq[lo:hi]).cuda(), "i8")
        m.set_weight("dn.scales", dev(ds[lo:hi]), "bf16")
        m.set_weight("dn.zeros", dev(dz[lo:hi]), "bf16")
        m.set_tensor("x", dev(x), "bf16")
        m.set_tensor("router", dev(logits), "bf16")
        op = m.create_op("MOEA16W8", "moe", ["x", "router"], ["y"], ["gu", "gu.scales", "gu.zeros", "dn", "dn.scales", "dn.zeros"],
                         f"num_experts=i:{E};num_experts_per_tok=i:{k};use_ep=i:1")
        m.reshape(op)
        m.alloc(op)
        m.forward(op)
        _, shape, ptr = m.get_tensor("y")
        total += view_of(ptr, shape, torch.bfloat16).float().cpu().numpy().reshape(T, hidden)
        if rank == 0:
            # CalcExpert on this rank's output
            w = bf16_round(rng.uniform(0, 1, (T, 1, 1)).astype(np.float32))
            m.set_tensor("w", dev(w), "bf16")
            ce = m.create_op("CalcExpert", "calc", ["y", "w"], ["c"], [], "num_experts=f:1")
            m.reshape(ce)
            m.alloc(ce)
            m.forward(ce)
            _, cshape, cptr = m.get_tensor("c")
            got = view_of(cptr, cshape, torch.bfloat16).float().cpu().numpy().reshape(T, hidden)
            y0 = view_of(ptr, shape, torch.bfloat16).float().cpu().numpy().reshape(T, hidden)
            np.testing.assert_array_equal(got, bf16_round(y0 * w.reshape(T, 1)))
        m.close()
    s_ref, e_ref = moe.route(logits.reshape(T, E), k)
    ref = moe.experts_ffn(x.reshape(T, hidden), e_ref, s_ref, list(zip(gq, gs, gz)), list(zip(uq, us, uz)), list(zip(dq, ds, dz)), G, 8)
    np.testing.assert_allclose(total, ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())


def test_row_split_gemm_adds_the_residual_on_rank0_only(env):
    """Tensor parallel o_proj / down_proj (HSPLIT) with the fused binary ADD: the AllReduce sums the ranks' outputs, so the
    residual must enter once -- GemmOpBase::Reshape drops it on rank != 0 (gemm_op.cpp:133-137).  Two ranks' operators over
    the two K halves: out_0 + out_1 == x.W + residual (not + 2 residual)."""
    hostapi, ops = env
    rng = np.random.default_rng(21)
    K, N, M, G = 512, 256, 3, 128
    W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
    q, s, z = quant.iq_quantize_a16w4(W, G, "bf16")
    x = bf16_round(rng.normal(0, 1, (M, 1, K)).astype(np.float32))
    res = bf16_round(rng.normal(0, 1, (M, 1, N)).astype(np.float32))
    outs = []
    for rank in range(2):
        rows = slice(rank * K // 2, (rank + 1) * K // 2)
        grp = slice(rank * K // 2 // G, (rank + 1) * K // 2 // G)
        m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16, rank=rank, nranks=2)
        m.set_weight("w", torch.from_numpy(np.ascontiguousarray(q[rows])).cuda(), "u8")
        m.set_weight("w.scales", dev(s[grp]), "bf16")
        m.set_weight("w.zeros", dev(z[grp]), "bf16")
        m.set_tensor("x", dev(x[:, :, rows]), "bf16")
        m.set_tensor("res", dev(res), "bf16")
        op = m.create_op("GemmA16W4", "decoder.layer.0.attention.output", ["x", "res"], ["y"], ["w", "w.scales", "w.zeros"],
                         f"alpha=f:1.0;GroupSize=i:{G}")
        m.reshape(op)
        m.alloc(op)
        m.forward(op)
        dt, shape, ptr = m.get_tensor("y")
        outs.append(view_of(ptr, shape, torch.bfloat16).float().cpu().numpy().reshape(M, N))
        m.close()
    full = gemm_ref.gemm_a16wx(x.reshape(M, K), q, s, z, G, 4, ft="f32", round_out=False)
    want = full + res.reshape(M, N)
    got = outs[0] + outs[1]
    tol = 3 * 2 ** -8 * np.abs(want).max()
    assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
    assert np.abs(got - (want + res.reshape(M, N))).max() > 10 * tol  # the bug this guards against: the residual summed twice


@pytest.mark.parametrize("rank,want_n", [(0, 4), (1, 3), (6, 4), (7, 3)])
def test_span_attention_operator_with_replicated_kv_heads(env, rank, want_n):
    """Qwen2-7B at TP = 8 (28 query / 4 KV heads): fewer KV heads than ranks -- each KV head lives on two ranks which take 4 and
    3 of its query heads (dash-infer_amd/tp.py::shard_heads; the reference refuses, head_gqa.h:29-49).  The operator derives
    the rank's head counts by the same rule: Reshape accepts the rank's fused qkv width and decode runs."""
    from dash_infer_amd import tp
    hostapi, ops = env
    shard = tp.shard_heads(28, 4, 8)[rank]
    assert len(shard.q_heads) == want_n and len(shard.kv_heads) == 1
    rng = np.random.default_rng(rank)
    n, g, H, S, L = want_n, 1, 128, 16, 21
    m = hostapi.Model(ops.cur_stream(), 28, 4, H, S, cache_mode=0, max_batch=1, max_len=64, rank=rank, nranks=8)
    pool = ops.SpanPool(16, g, S, H, "none", torch.bfloat16)
    kv = ops.KVCacheSet(pool, 1, 4)
    kv.ensure(0, L + 1)
    kc, vc = kv_codec.SpanCache(g, S, H, "none"), kv_codec.SpanCache(g, S, H, "none")
    for t in range(L):
        kc.write(t, rng.normal(0, 1, (g, H)))
        vc.write(t, rng.normal(0, 1, (g, H)))
    for i, sp in enumerate(kc.spans):
        pool.span_view(kv.k_idx[0][i]).copy_(torch.from_numpy(sp))
    for i, sp in enumerate(vc.spans):
        pool.span_view(kv.v_idx[0][i]).copy_(torch.from_numpy(sp))
    qkv = bf16_round(rng.normal(0, 1, (1, 1, (n + 2 * g) * H)).astype(np.float32))
    m.set_tensor("qkv", dev(qkv), "bf16")
    op = m.create_op("DecOptMQA", "decoder.layer.0.attention", ["qkv"], ["out"])
    kptrs = [[[int(kv.k_host[0, i]) for i in range(kv.max_spans)]]]
    vptrs = [[[int(kv.v_host[0, i]) for i in range(kv.max_spans)]]]
    m.set_runtime(False, [L], kptrs, vptrs)
    m.reshape(op)
    m.alloc(op)
    m.forward(op)
    dt, shape, ptr = m.get_tensor("out")
    assert shape == [1, 1, n * H]
    out = view_of(ptr, shape, torch.bfloat16).float().cpu().numpy().reshape(n, H)
    kc.write(L, qkv[0, 0, n * H:(n + g) * H].reshape(g, H))
    vc.write(L, qkv[0, 0, (n + g) * H:].reshape(g, H))
    ref = attention.decode_attention(qkv[0, 0, : n * H].reshape(n, H), kc.read_all(L + 1), vc.read_all(L + 1), 1.0 / np.sqrt(H))
    np.testing.assert_allclose(out, ref, rtol=1e-2, atol=2.5e-3)
    # a width that belongs to another rank's share is rejected
    m.set_tensor("qkv", dev(np.zeros((1, 1, ((7 - want_n) + 2) * H), np.float32)), "bf16")
    with pytest.raises(hostapi.HostError):
        m.reshape(op)
    m.close()


def test_span_attention_alloc_claims_spans_through_the_virtual_cache(env):
    """SpanAttnOp::Alloc (span_attn_op.cpp:315-368): every step each request's cache grows by the tokens the step appends,
    through VirtualCache::GetCache(layer, increment); the sanity check step == cached length is the reference's; running out
    of spans surfaces as ALLSPARK_CACHE_MEMORY_OUT; the Alloc of different layers may run concurrently (model.cpp:1253-1262)."""
    import ctypes as C
    hostapi, ops = env
    n, g, H, S, layers, max_len = 4, 2, 128, 16, 6, 64
    spr = max_len // S
    m = hostapi.Model(ops.cur_stream(), n, g, H, S, 0, max_batch=2, max_len=max_len)
    pool = ops.SpanPool(2 * 2 * layers * spr + 1, g, S, H, "none", torch.bfloat16)
    kp = [[[pool.alloc()[0] for _ in range(spr)] for _ in range(layers)] for _ in range(2)]
    vp_ = [[[pool.alloc()[0] for _ in range(spr)] for _ in range(layers)] for _ in range(2)]
    m.set_tensor("qkv", torch.zeros(2, 1, (n + 2 * g) * H, dtype=torch.bfloat16, device="cuda"), "bf16")
    opids = [m.create_op("DecOptMQA", f"decoder.layer.{l}.attention", ["qkv"], [f"out{l}"]) for l in range(layers)]
    steps = [15, 31]   # both requests are about to cross a span boundary
    m.set_runtime(False, steps, kp, vp_)
    for o in opids:
        m.reshape(o)
    lib = hostapi.lib()
    assert lib.dihost_ops_alloc_concurrent(m.h, (C.c_int * layers)(*opids), layers) == 0
    for l in range(layers):
        assert lib.dihost_cache_seq_len(m.h, 0, l) == 16 and lib.dihost_cache_seq_len(m.h, 1, l) == 32
    for o in opids:
        m.forward(o)   # decode attention over the claimed spans (zeros: only checks that every span was there)
    torch.cuda.synchronize()
    # a second Alloc without advancing gen_ctx->step: the reference's sanity check (cached length != step)
    with pytest.raises(hostapi.HostError) as e:
        m.alloc(opids[0])
    assert e.value.code == 5
    # out of spans: a request at the end of its last span
    m.set_runtime(False, [max_len, 3], kp, vp_)
    with pytest.raises(hostapi.HostError) as e:
        m.alloc(opids[1])
    assert e.value.code == 11  # ALLSPARK_CACHE_MEMORY_OUT
    m.close()


def test_allgather_operator_single_rank_and_row_transpose(env):
    """AllGatherOp (allgather_op.cpp:27-58,134-163): out = [rows, nranks * n]; one rank = copy.  The rank-major -> row-major
    transpose behind ncclAllGather is exercised through the C-ABI with a one-rank 'gathered' buffer built by hand."""
    hostapi, ops = env
    from dash_infer_amd.capi import check, lib
    m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16)
    x = torch.randn(3, 1, 40, device="cuda").to(torch.bfloat16)
    m.set_tensor("x", x, "bf16")
    op = m.create_op("AllGather", "embedding.allgather", ["x"], ["y"])
    m.reshape(op)
    m.forward(op)
    dt, shape, ptr = m.get_tensor("y")
    assert shape == [3, 1, 40] and torch.equal(view_of(ptr, shape, torch.bfloat16), x)
    m.close()
    # the transpose behind the collective: "ranks" x rows x row bytes as ncclAllGather leaves them, 16- / 4- / 1-byte paths
    for nr, rows, rb in ((4, 5, 24), (8, 3, 256), (2, 7, 5), (8, 1, 7168)):
        tmp = torch.randint(0, 256, (nr * rows * rb,), dtype=torch.uint8, device="cuda")
        out = torch.empty_like(tmp)
        check(lib().dihip_gather_rows_transpose(ops.cur_stream(), ops.ptr(out), ops.ptr(tmp), nr, rows, rb), "gather_rows_transpose")
        torch.cuda.synchronize()
        assert torch.equal(out, tmp.view(nr, rows, rb).transpose(0, 1).reshape(-1)), (nr, rows, rb)


def test_preprocess_id_and_update_id_follow_a_request_through_its_stop_conditions(env):
    """PreProcessId (preprocess_id_op.cpp:32-80) copies the request's input ids into its host and device "generated_ids"; after
    every step GenerateOp writes the new token there (fill_generated_ids) and UpdateId (update_id_op.cpp:42-156) queues it and
    applies the stop conditions: eos (early_stopping), stop words, the length limit -- positions per the reference's table
    (context: step + in_length_bias, decoder: step)."""
    from dash_infer_amd import hostapi, ops
    m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16, max_batch=3, max_len=64)
    pre = m.create_op("PreProcessId", "preprocess_id", ["input_ids"], ["pre.out"])
    upd = m.create_op("UpdateId", "update_id", ["generated_ids"], ["upd.out"])
    prompt = [101, 102, 103, 104]
    L = len(prompt)
    scripts = {   # tokens GenerateOp "samples" per step (context step first) and the step after which the request must be finished
        "eos":   dict(toks=[5, 6, 99, 7], kw=dict(max_length=64, eos=99, stop_words=[]), done_after=2),
        "words": dict(toks=[5, 7, 8, 9], kw=dict(max_length=64, eos=-1, stop_words=[[7, 8], [1, 2]]), done_after=2),
        "limit": dict(toks=[5, 6, 7, 8], kw=dict(max_length=L + 3, eos=-1, stop_words=[]), done_after=2),
    }
    for name, sc in scripts.items():
        # ---- context phase: step = 0 (no prefix); PreProcessId; GenerateOp's token lands at position L; UpdateId with in_length_bias = L
        m.set_runtime(True, [0], [[[]]], [[[]]])
        m.request_attach(0, prompt, early_stopping=True, in_length_bias=0, **sc["kw"])
        m.forward(pre)
        _, _, n_interim = m.request_poll(0)
        assert n_interim == 2                                 # generated_ids + generated_ids_gpu
        m.request_set_step(0, 0, in_length_bias=L)
        m.request_put_token(0, L, sc["toks"][0])
        m.forward(upd)
        got, finish, _ = m.request_poll(0)
        assert got == [sc["toks"][0]] and not finish, name
        # ---- decoder phase: step = tokens in the cache = position of the token fed; the new token lands at step + 1 ... the
        # reference reads generated_ids[step] AFTER AsModel advanced step (model.cpp:1320): emulate that order
        m.set_phase(False)
        for t, tok in enumerate(sc["toks"][1:], start=1):
            m.request_put_token(0, L + t, tok)
            m.request_set_step(0, L + t, in_length_bias=0)
            m.forward(upd)
            got, finish, _ = m.request_poll(0)
            assert got == [tok], (name, t)
            assert finish == (t >= sc["done_after"]), (name, t, finish)
    m.close()


def test_weights_from_a_serialized_file(env):
    """A model whose weights come from a serialized weight file (tests/golden/tiny_qwen2_a16w4.asparam: written by the REFERENCE'S OWN
    writer, tests/golden/make_asparam_golden.py) through `dihost_weights_load_file` (host/weight_file.h): every record sits in device memory
    under its name with the bytes of the file, and the fused operator list built over it -- context phase and decode steps -- gives logits
    BIT-IDENTICAL to the same list over the same arrays bound from torch tensors (the path every other test takes).  Together with
    tests/test_host_graph_serialized.py this is the reference's export -- graph + weights, as AsModel reads them (model.cpp:265-287,
    weight_manager.cpp) -- entering the C++ layer with nothing hand-built in between."""
    import importlib.util
    import os
    hostapi, ops = env
    from dash_infer_amd import ref_graph
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", "tiny_qwen2_a16w4.asparam")
    spec = importlib.util.spec_from_file_location("make_asparam_golden", os.path.join(root, "tests", "golden", "make_asparam_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    recs = mod.tiny_model()
    hidden, n, g, H, vocab, span, max_len = 256, 2, 1, 128, 320, 16, 64
    TORCH = {"float32": torch.float32, "uint8": torch.uint8, "int64": torch.int64}
    NAME = {"float32": "f32", "uint8": "u8", "int64": "i64"}

    def run(from_file):
        stream = torch.cuda.Stream()
        torch.cuda.synchronize()
        pool = ops.SpanPool(2 * 1 * (max_len // span) + 4, g, span, H, "none", torch.bfloat16)
        keep = []
        with torch.cuda.stream(stream):
            m = hostapi.Model(ops.cur_stream(), n, g, H, span, 0, max_batch=1, max_len=max_len)
            if from_file:
                assert m.load_weight_file(path) == len(recs)
                for name, arr, _, bf16 in recs:      # the bytes in device memory are the file's
                    dt, shape, ptr = m.get_weight(name)
                    assert shape == list(arr.shape), name
                    got = view_of(ptr, [int(arr.nbytes)], torch.uint8).cpu().numpy()
                    assert got.tobytes() == np.ascontiguousarray(arr).tobytes(), name
            else:
                for name, arr, _, bf16 in recs:
                    if bf16:
                        t = torch.from_numpy(arr.view(np.int16).copy()).cuda().view(torch.bfloat16)
                        m.set_weight(name, t, "bf16")
                    else:
                        t = torch.from_numpy(arr.copy()).cuda()
                        m.set_weight(name, t, NAME[str(arr.dtype)])
                    keep.append(t)
            ref_graph.add_graph(m, ref_graph.qwen2_graph(1, 4, 128, 1e-6, n, g, 1e6))
            rep = m.graph_build(fuse=True)
            assert rep["fused"], rep["why"]
            if from_file:   # the operators released the sources they re-laid out: get_weight says so instead of handing out a null pointer as data
                released = 0
                for name, *_ in recs:
                    try:
                        m.get_weight(name)
                    except hostapi.HostError as e:
                        assert e.code == 8 and "released" in str(e), str(e)
                        released += 1
                assert released >= 4
            ks = [[pool.alloc()[0] for _ in range(max_len // span)]]
            vs = [[pool.alloc()[0] for _ in range(max_len // span)]]
            prompt = [int(t) for t in np.random.default_rng(2).integers(0, vocab, 21)]
            out = [m.request_start(prompt, ks, vs)]
            logits = []
            for _ in range(3):
                m.decode_steps(1, graph=True)
                out += m.sync_ids()
                _, shp, ptr = m.get_tensor("logits")
                stream.synchronize()
                logits.append(view_of(ptr, shp, torch.float32).clone())
            m.close()
        return out, logits

    ids_a, lo_a = run(True)
    ids_b, lo_b = run(False)
    assert ids_a == ids_b
    for t, (x, y) in enumerate(zip(lo_a, lo_b)):
        assert torch.equal(x, y), f"step {t}: logits differ between file-loaded and tensor-bound weights"
        assert torch.isfinite(x).all()
