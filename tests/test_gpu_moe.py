"""GPU parity of the mixture-of-experts decode path (dihip_moe_route / dihip_moe_experts, C-ABI) against oracle/moe.py.

Cases: a small expert stack for every weight format, ragged routing (ties, an expert hit by several tokens, an
expert nobody picks), skipped slots (expert parallelism: index -1), and the Qwen2-57B-A14B shape of
BASELINE configs[4] (64 experts, top-8, hidden 3584, expert width 2560) with oracle spot checks plus size-independent
properties: run-to-run determinism and linearity of the combine in the routing weights.
"""
import numpy as np
import pytest
import torch

from oracle import moe, quant
from oracle.numerics import bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(pkg):
    from dash_infer_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dt) if dt is not None else t).cuda()


def make_experts(rng, E, N, K, G, wbits):
    qs, ss, zs = [], [], []
    for _ in range(E):
        W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
        q, s, z = (quant.iq_quantize_a16w8 if wbits == 8 else quant.iq_quantize_a16w4)(W, G, "bf16")
        qs.append(q)
        ss.append(s)
        zs.append(z)
    return qs, ss, zs


def pack(ops, qs, ss, zs, G, wbits):
    return ops.pack_experts([dev(q) for q in qs], [dev(s, torch.bfloat16) for s in ss], [dev(z, torch.bfloat16) for z in zs], G, wbits)


@pytest.mark.parametrize("T,E,k", [(1, 8, 2), (5, 60, 4), (3, 64, 8), (2, 200, 6)])
def test_route_matches_oracle(ops, T, E, k):
    rng = np.random.default_rng(E + k)
    logits = bf16_round(rng.normal(0, 2, (T, E)).astype(np.float32))
    logits[0, 1] = logits[0, 3] = logits[0].max() + 1.0   # an exact tie at the top: lower index first
    s_ref, e_ref = moe.route(logits, k)
    for dt in (torch.bfloat16, torch.float32):
        s, e = ops.moe_route(dev(logits, dt), k)
        np.testing.assert_array_equal(e.cpu().numpy(), e_ref)
        np.testing.assert_allclose(s.cpu().numpy(), s_ref, rtol=2e-6, atol=1e-9)
    assert e_ref[0, 0] == 1 and e_ref[0, 1] == 3


@pytest.mark.parametrize("wbits,G", [(8, -1), (8, 128), (4, 128)])
def test_experts_small_matches_oracle(ops, wbits, G):
    rng = np.random.default_rng(wbits * 10 + (G > 0))
    T, E, k, hidden, proj = 4, 6, 3, 256, 384
    x = bf16_round(rng.normal(0, 1, (T, hidden)).astype(np.float32))
    gate, up, down = (make_experts(rng, E, proj, hidden, G, wbits), make_experts(rng, E, proj, hidden, G, wbits),
                      make_experts(rng, E, hidden, proj, G, wbits))
    pg, pu, pd = pack(ops, *gate, G, wbits), pack(ops, *up, G, wbits), pack(ops, *down, G, wbits)
    logits = bf16_round(rng.normal(0, 1.5, (T, E)).astype(np.float32))
    logits[:, 5] = -30.0            # expert 5 is never picked; expert 0 by everyone
    logits[:, 0] = 4.0
    scores, experts = ops.moe_route(dev(logits, torch.bfloat16), k)
    s_ref, e_ref = moe.route(logits, k)
    np.testing.assert_array_equal(experts.cpu().numpy(), e_ref)
    out = ops.moe_experts(dev(x, torch.bfloat16), experts, scores, pg, pu, pd)
    ref = moe.experts_ffn(x, e_ref, s_ref, list(zip(*gate)), list(zip(*up)), list(zip(*down)), G, wbits)
    o = out.float().cpu().numpy()
    np.testing.assert_allclose(o, ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())   # bf16 intermediates (two roundings)
    # expert parallelism: slots whose expert lives on another rank (-1) contribute nothing
    ex2 = experts.clone()
    ex2[:, 1] = -1
    out2 = ops.moe_experts(dev(x, torch.bfloat16), ex2, scores, pg, pu, pd)
    e2 = e_ref.copy()
    e2[:, 1] = -1
    ref2 = moe.experts_ffn(x, e2, s_ref, list(zip(*gate)), list(zip(*up)), list(zip(*down)), G, wbits)
    np.testing.assert_allclose(out2.float().cpu().numpy(), ref2, rtol=2e-2, atol=2e-2 * np.abs(ref2).max())


def test_experts_qwen2_57b_a14b_shape(ops):
    """configs[4] expert shapes at int8 per-channel: 64 experts x (3584 -> 2560 gate/up, 2560 -> 3584 down), top-8."""
    rng = np.random.default_rng(57)
    T, E, k, hidden, proj, wbits, G = 2, 64, 8, 3584, 2560, 8, -1
    # one random expert triple replicated with per-expert scale tweaks keeps host-side quantisation time small
    base = make_experts(rng, 3, proj, hidden, G, wbits), make_experts(rng, 3, proj, hidden, G, wbits), make_experts(rng, 3, hidden, proj, G, wbits)
    def stack(trip):
        qs, ss, zs = trip
        return ([qs[e % 3] for e in range(E)], [bf16_round(ss[e % 3] * (1.0 + 0.01 * (e // 3))) for e in range(E)],
                [zs[e % 3] for e in range(E)])
    gate, up, down = stack(base[0]), stack(base[1]), stack(base[2])
    pg, pu, pd = pack(ops, *gate, G, wbits), pack(ops, *up, G, wbits), pack(ops, *down, G, wbits)
    x = bf16_round(rng.normal(0, 1, (T, hidden)).astype(np.float32))
    logits = bf16_round(rng.normal(0, 1, (T, E)).astype(np.float32))
    scores, experts = ops.moe_route(dev(logits, torch.bfloat16), k)
    xd = dev(x, torch.bfloat16)
    out = ops.moe_experts(xd, experts, scores, pg, pu, pd)
    assert torch.equal(out, ops.moe_experts(xd, experts, scores, pg, pu, pd))            # deterministic
    # combine is linear in the routing weights: doubling them doubles the (pre-rounding) result exactly
    out2 = ops.moe_experts(xd, experts, scores * 2, pg, pu, pd)
    assert torch.equal(out2, out * 2)
    # oracle on token 0 restricted to its first 2 experts (the oracle's python loops are slow at this size)
    e_np, s_np = experts.cpu().numpy(), scores.cpu().numpy()
    e1 = np.full_like(e_np, -1)
    e1[0, :2] = e_np[0, :2]
    part = ops.moe_experts(xd, dev(e1), scores, pg, pu, pd).float().cpu().numpy()
    ref = moe.experts_ffn(x, e1, s_np, list(zip(*gate)), list(zip(*up)), list(zip(*down)), G, wbits)
    np.testing.assert_allclose(part, ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())
    assert np.all(part[1] == 0)


@pytest.mark.parametrize("wbits,G,nranks", [(8, -1, 8), (4, 128, 4)])
def test_experts_tensor_parallel_split(ops, wbits, G, nranks):
    """configs[4] runs the experts under TP: every rank holds a column slice of each expert's gate / up and the matching
    row slice of its down projection (the FFN split of SURVEY 8(e): whole quantisation groups per rank, per-channel
    scales of the row-split matrix not split), calls the SAME entry point with its local width, and the outputs are
    summed by the all-reduce.  SwiGLU is column-wise and the down projection linear, so the sum over ranks must equal
    the unsplit call (to the rounding of the per-rank bf16 outputs)."""
    rng = np.random.default_rng(7 * nranks + wbits)
    T, E, k, hidden = 3, 8, 3, 256
    proj = 64 * nranks if wbits == 8 else 128 * nranks
    x = bf16_round(rng.normal(0, 1, (T, hidden)).astype(np.float32))
    gate, up, down = (make_experts(rng, E, proj, hidden, G, wbits), make_experts(rng, E, proj, hidden, G, wbits),
                      make_experts(rng, E, hidden, proj, G, wbits))
    logits = bf16_round(rng.normal(0, 1.5, (T, E)).astype(np.float32))
    scores, experts = ops.moe_route(dev(logits, torch.bfloat16), k)
    xd = dev(x, torch.bfloat16)
    full = ops.moe_experts(xd, experts, scores, pack(ops, *gate, G, wbits), pack(ops, *up, G, wbits), pack(ops, *down, G, wbits))

    def cols(trip, lo, hi):   # column slice of [K, N] weights (uint4: two columns per byte) and of their [G, N] parameters
        qs, ss, zs = trip
        if wbits == 4:
            qs = [quant.pack_u4(quant.unpack_u4(q, proj)[:, lo:hi]) for q in qs]
        else:
            qs = [q[:, lo:hi] for q in qs]
        return qs, [s_[:, lo:hi] for s_ in ss], [z[:, lo:hi] for z in zs]

    def rows(trip, lo, hi):   # row (K) slice: whole groups; per-channel parameters stay whole
        qs, ss, zs = trip
        qs = [q[lo:hi] for q in qs]
        if G > 0:
            ss, zs = [s_[lo // G:hi // G] for s_ in ss], [z[lo // G:hi // G] for z in zs]
        return qs, ss, zs

    per = proj // nranks
    acc = np.zeros((T, hidden), np.float64)
    for r in range(nranks):
        lo, hi = r * per, (r + 1) * per
        out_r = ops.moe_experts(xd, experts, scores, pack(ops, *cols(gate, lo, hi), G, wbits), pack(ops, *cols(up, lo, hi), G, wbits),
                                pack(ops, *rows(down, lo, hi), G, wbits))
        acc += out_r.float().cpu().numpy()
    ref = full.float().cpu().numpy()
    np.testing.assert_allclose(acc, ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())


@pytest.mark.parametrize("wbits,G,T,E,k", [(8, -1, 16, 4, 2), (4, 128, 9, 6, 3), (8, 128, 5, 5, 1), (8, -1, 32, 64, 8)])
def test_grouped_experts_equal_the_per_token_calls(ops, wbits, G, T, E, k):
    """More than one token: slots that picked the same expert are gathered into groups of up to 4 rows that stream the expert
    once (moe_group_kernel + the MR = 4 slot GEMV).  A one-token call takes the per-slot launches: row t of the T-token call
    must equal the call on token t alone up to the f32 summation order of the K split (the launch plan -- units per
    workgroup, waves across K -- follows the slot count, so the two calls may split K differently) -- with few experts
    (every expert hit by many tokens: full groups, partial last groups), with skipped slots (expert parallelism: -1), and
    with every slot's expert distinct per token (top-k of one token never repeats an expert)."""
    rng = np.random.default_rng(T * 7 + E + k)
    hidden, proj = 256, 384
    gate = pack(ops, *make_experts(rng, E, proj, hidden, G, wbits), G, wbits)
    up = pack(ops, *make_experts(rng, E, proj, hidden, G, wbits), G, wbits)
    down = pack(ops, *make_experts(rng, E, hidden, proj, G, wbits), G, wbits)
    x = dev(bf16_round(rng.normal(0, 1, (T, hidden)).astype(np.float32)), torch.bfloat16)
    logits = dev(bf16_round(rng.normal(0, 2, (T, E)).astype(np.float32)), torch.bfloat16)
    scores, experts = ops.moe_route(logits, k)
    for ep in (None, (0, max(1, E // 2))):          # whole stack / expert-parallel window (slots outside it are skipped)
        if ep is not None:
            scores, experts = ops.moe_route(logits, k, ep=ep)
            if int((experts >= 0).sum()) == 0:
                continue
            # the window's experts are the first E/2 of the same stacks: positions coincide with global ids
        whole = ops.moe_experts(x, experts, scores, gate, up, down)
        torch.cuda.synchronize()
        for t in range(T):
            one = ops.moe_experts(x[t:t + 1].contiguous(), experts[t:t + 1].contiguous(), scores[t:t + 1].contiguous(), gate, up, down)
            torch.cuda.synchronize()
            a, b = whole[t].float(), one[0].float()
            tol = 2.0 ** -7 * max(float(b.abs().max()), 1e-3)   # two bf16 roundings (expert output, combine) of sums in another order
            assert float((a - b).abs().max()) <= tol, f"token {t} (ep={ep}): max diff {(a - b).abs().max()} > {tol}"
    # run-to-run determinism of the grouped path
    again = ops.moe_experts(x, experts, scores, gate, up, down)
    torch.cuda.synchronize()
    assert torch.equal(again, whole)


@pytest.mark.parametrize("wbits,G,T,E,k", [(8, -1, 16, 64, 8), (4, 128, 9, 6, 3), (8, 128, 2, 5, 1), (8, -1, 33, 64, 8)])
def test_fewer_launch_block_is_bit_identical(ops, wbits, G, T, E, k):
    """dihip_moe_route_grouped + dihip_moe_experts_ex(PREGROUPED | NO_FINALIZE) + dihip_moe_combine against the separate calls
    (route, experts with its own grouping launch and finalize kernel, shared_combine): the same scores / expert ids, the same
    group tables, and the same f32 hidden rows bit for bit -- whole stack and an expert-parallel window, with and without the
    residual rows (ranks > 0 pass none)."""
    rng = np.random.default_rng(T * 11 + E + k)
    hidden, proj = 256, 384
    gate = pack(ops, *make_experts(rng, E, proj, hidden, G, wbits), G, wbits)
    up = pack(ops, *make_experts(rng, E, proj, hidden, G, wbits), G, wbits)
    down = pack(ops, *make_experts(rng, E, hidden, proj, G, wbits), G, wbits)
    x = dev(bf16_round(rng.normal(0, 1, (T, hidden)).astype(np.float32)), torch.bfloat16)
    logits = dev(bf16_round(rng.normal(0, 2, (T, E)).astype(np.float32)), torch.bfloat16)
    shared = dev(bf16_round(rng.normal(0, 1, (T, hidden)).astype(np.float32)), torch.bfloat16)
    sig = dev(bf16_round(rng.uniform(0, 1, (T, 1)).astype(np.float32)), torch.bfloat16)
    h = dev(rng.normal(0, 1, (T, hidden)).astype(np.float32), torch.float32)
    nbytes = int(ops.lib().dihip_moe_workspace_bytes(T, k, hidden, proj))
    for ep in (None, (0, max(1, E // 2))):
        for with_res in (True, False):
            s0, e0 = ops.moe_route(logits, k, ep=ep)
            moe_out = ops.moe_experts(x, e0, s0, gate, up, down)
            ref = torch.empty_like(h)
            ops.moe_shared_combine(ref, h if with_res else None, moe_out, shared, sig)
            ws = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")
            s1, e1 = ops.moe_route_grouped(logits, k, hidden, proj, ws, ep=ep)
            assert torch.equal(s0, s1) and torch.equal(e0, e1)
            assert ops.moe_experts(x, e1, s1, gate, up, down, ws=ws, flags=ops.MOE_PREGROUPED | ops.MOE_NO_FINALIZE) is None
            got = torch.empty_like(h)
            ops.moe_combine(got, h if with_res else None, ws, s1, e1, shared, sig, proj)
            torch.cuda.synchronize()
            assert torch.equal(got, ref), f"ep={ep} residual={with_res}: max diff {(got - ref).abs().max()}"


def test_fewer_launch_block_rejects_what_it_does_not_serve(ops):
    logits = torch.zeros(1, 8, dtype=torch.bfloat16, device="cuda")
    ws = torch.zeros(int(ops.lib().dihip_moe_workspace_bytes(300, 8, 64, 64)), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        ops.moe_route_grouped(logits, 2, 64, 64, ws)                       # one token: its top-k experts are distinct, no groups
    with pytest.raises(Exception):
        ops.moe_route_grouped(torch.zeros(300, 8, dtype=torch.bfloat16, device="cuda"), 8, 64, 64, ws)   # 2400 slots > 2048


@pytest.mark.parametrize("T,E,hidden,dt", [(16, 64, 3584, torch.bfloat16), (1, 64, 3584, torch.bfloat16), (33, 60, 512, torch.bfloat16),
                                           (7, 8, 256, torch.float16)])
def test_router_and_gate_in_one_launch(ops, T, E, hidden, dt):
    """dihip_moe_router_gate against the two Gemm operators it replaces (dihip_gemm_a16w16, the second with SIGMOID) and against
    float64: the same products summed in another fixed order (16 waves x k-steps, then 16 partial tiles) -- within one FT
    rounding of each other, run-to-run identical."""
    rng = np.random.default_rng(T * 3 + E)
    ft = "bf16" if dt == torch.bfloat16 else "f16"
    from oracle.numerics import ft_round
    xn = ft_round(rng.normal(0, 1, (T, hidden)).astype(np.float32), ft)
    wr = ft_round(rng.normal(0, 0.05, (hidden, E)).astype(np.float32), ft)
    wg = ft_round(rng.normal(0, 0.05, (hidden, 1)).astype(np.float32), ft)
    xd = dev(xn, dt)
    pr, pg = ops.pack_dense(dev(wr, dt)), ops.pack_dense(dev(wg, dt))
    logits, sig = ops.moe_router_gate(xd, pr, pg)
    ref_l = ops.gemm_dense(xd, pr)
    ref_s = ops.gemm_dense(xd, pg, act="sigmoid")
    torch.cuda.synchronize()
    exact_l = xn.astype(np.float64) @ wr.astype(np.float64)
    exact_s = 1.0 / (1.0 + np.exp(-(xn.astype(np.float64) @ wg.astype(np.float64))))
    ulp = 2.0 ** -8 if ft == "bf16" else 2.0 ** -11
    assert np.abs(logits.float().cpu().numpy() - exact_l).max() <= ulp * max(1.0, np.abs(exact_l).max())
    assert np.abs(sig.float().cpu().numpy() - exact_s).max() <= ulp
    assert float((logits.float() - ref_l.float()).abs().max()) <= 2 * ulp * max(1.0, float(ref_l.float().abs().max()))
    assert float((sig.float() - ref_s.float()).abs().max()) <= 2 * ulp
    l2, s2 = ops.moe_router_gate(xd, pr, pg)
    torch.cuda.synchronize()
    assert torch.equal(l2, logits) and torch.equal(s2, sig)
