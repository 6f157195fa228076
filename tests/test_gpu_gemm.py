"""GPU parity of the weight-only GEMM/GEMV (through the C-ABI) against the oracle.

Cases follow the reference's operator tests (tests/cpp/operator/cuda/operator_gemm_lowp_test.cpp
:1034-1115: M in {1,3,17,31,128,..}, ragged N/K, group 64/128/256, alpha != 1) at sizes the
oracle finishes in seconds, the BASELINE.json layer shapes of Qwen2-7B at M=1 against the C
oracle directly, and size-independent properties at full size (batch invariance, run-to-run
determinism of the split-K reduction, linearity).

Tolerance: the kernel accumulates exact products in f32 and rounds once to FT; the oracle sums in
f64 and rounds once.  Allowed: one FT ulp (2^-7 relative for bf16, 2^-10 for f16) plus a small
absolute term for cancellation -- far inside the reference's own 1e-1 (A16W8) bound
(operator_gemm_lowp_test.cpp:470-471), which is also asserted with its check_equal metric.
"""
import numpy as np
import pytest
import torch

from oracle import cbind, gemm_ref, glue, quant
from oracle.numerics import bf16_round, check_equal, f16_round
from tests import helpers

pytestmark = pytest.mark.gpu

FT = {"bf16": torch.bfloat16, "f16": torch.float16}
ULP = {"bf16": 2.0 ** -7, "f16": 2.0 ** -10}


@pytest.fixture(scope="module")
def ops(pkg):
    from dash_infer_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def rnd(ft):
    return bf16_round if ft == "bf16" else f16_round


def to_dev(a, ft=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if ft is not None:
        t = t.to(FT[ft])
    return t.cuda()


def make_case(rng, M, N, K, G, wbits, ft, style="iq"):
    r = rnd(ft)
    x = r(rng.uniform(-1, 1, (M, K)).astype(np.float32))
    if style == "iq":  # InstantQuant of a random weight (quantization_utils.py)
        W = r(rng.normal(0, 0.02, (K, N)).astype(np.float32))
        q, s, z = (quant.iq_quantize_a16w8 if wbits == 8 else quant.iq_quantize_a16w4)(W, G, ft)
    else:  # the reference test's synthetic distribution (operator_gemm_lowp_test.cpp:486-489)
        Gn = (K + G - 1) // G if G > 0 else 1
        s = r(rng.uniform(0.9 * 2 / 256, 1.1 * 2 / 256, (Gn, N)).astype(np.float32))
        z = r(rng.uniform(-10, 10, (Gn, N)).astype(np.float32))
        if wbits == 8:
            q = rng.integers(-128, 128, (K, N)).astype(np.int8)
        else:
            q = quant.pack_u4(rng.integers(0, 16, (K, N)).astype(np.uint8))
    return x, q, s, z


def assert_close(out, ref, ft, scale_hint=None, what="", pre=None):
    """`pre`: an intermediate that was itself rounded to FT before the final op (residual add): a
    one-ulp flip of it is legitimate (libm tanhf/expf differ in the last bit between host and device)."""
    out = np.asarray(out, np.float64)
    ref = np.asarray(ref, np.float64)
    mag = np.abs(ref).max() if scale_hint is None else scale_hint
    tol = ULP[ft] * np.abs(ref) + ULP[ft] * 0.02 * max(mag, 1e-6) + 1e-6
    if pre is not None:
        tol = tol + ULP[ft] * np.abs(np.asarray(pre, np.float64))
    bad = np.abs(out - ref) > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} elements off, max err {np.abs(out - ref).max():.3e} (ref max {mag:.3e})"


# ------------------------------------------------------------------ layout (bit exact) -------
@pytest.mark.parametrize("wbits,K,N,G", [(4, 256, 64, 128), (4, 200, 50, 64), (8, 192, 48, -1), (8, 130, 33, 32),
                                         (4, 3584, 512, 128)])
def test_pack_kernels_byte_exact(ops, wbits, K, N, G):
    rng = np.random.default_rng(K + N)
    x, q, s, z = make_case(rng, 1, N, K, G, wbits, "bf16", style="synthetic")
    from oracle.numerics import bf16_bits
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
    torch.cuda.synchronize()
    spec_w = helpers.pack_tile_major(q, N, wbits)
    got_w = pw.w.cpu().numpy().view(np.uint32).reshape(spec_w.shape)
    np.testing.assert_array_equal(got_w, spec_w)
    spec_sz = helpers.pack_sz(bf16_bits(s), bf16_bits(z), N, K, G)
    got_sz = pw.sz.cpu().numpy().view(np.uint32).reshape(spec_sz.shape)
    np.testing.assert_array_equal(got_sz, spec_sz)


def test_pack_dense_byte_exact(ops):
    from oracle.numerics import bf16_bits
    rng = np.random.default_rng(1)
    K, N = 100, 40
    w = bf16_round(rng.normal(0, 1, (K, N)).astype(np.float32))
    pw = ops.pack_dense(to_dev(w, "bf16"))
    torch.cuda.synchronize()
    spec = helpers.pack_tile_major(bf16_bits(w), N, 16)
    np.testing.assert_array_equal(pw.w.cpu().numpy().view(np.uint32).reshape(spec.shape), spec)


# ------------------------------------------------------------------ operator-level parity ----
SMALL = [  # (M, N, K, G)
    (1, 256, 512, -1), (1, 256, 512, 128), (3, 320, 640, 64), (17, 261, 519, 128), (31, 512, 1024, 256),
    (1, 2560, 5120, 128), (2, 48, 96, 32), (16, 64, 128, -1), (33, 80, 256, 64), (128, 320, 512, 128),
]


@pytest.mark.parametrize("ft", ["bf16", "f16"])
@pytest.mark.parametrize("wbits", [8, 4])
@pytest.mark.parametrize("M,N,K,G", SMALL)
def test_gemm_a16wx_matches_oracle(ops, M, N, K, G, wbits, ft):
    rng = np.random.default_rng(M * 7 + N + K + wbits)
    x, q, s, z = make_case(rng, M, N, K, G, wbits, ft, style="iq")
    alpha = 0.75 if (M % 2) else 1.0
    ref = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, alpha=alpha, ft=ft)
    pw = ops.pack_lowp(to_dev(q), to_dev(s, ft), to_dev(z, ft), G, wbits)
    y = ops.gemm_lowp(to_dev(x, ft), pw, alpha=alpha)
    torch.cuda.synchronize()
    out = y.float().cpu().numpy()
    assert_close(out, ref, ft, what=f"M{M} N{N} K{K} G{G} w{wbits} {ft}")
    assert check_equal(ref, out) <= 1e-1  # the reference's own criterion


@pytest.mark.parametrize("wbits", [8, 4])
def test_reference_test_distribution(ops, wbits):
    """operator_gemm_lowp_test.cpp:486-489: q small ints, scale ~ 2/256, zero ~ U(-10,10) (fractional)."""
    rng = np.random.default_rng(42)
    M, N, K, G = 3, 512, 1024, 128
    x, q, s, z = make_case(rng, M, N, K, G, wbits, "bf16", style="synthetic")
    ref = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, ft="bf16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
    y = ops.gemm_lowp(to_dev(x, "bf16"), pw)
    out = y.float().cpu().numpy()
    assert_close(out, ref, "bf16", what="synthetic")
    if cbind.reflib() is not None and wbits == 8:
        theirs = cbind.ref_gemm_a16w8(x, q, s, z, G, 1.0, "bf16")  # the reference's own host loop
        assert check_equal(theirs, out) <= 1e-1


@pytest.mark.parametrize("act", [None, "silu", "relu", "gelu_erf", "gelu_tanh", "tanh", "sigmoid"])
def test_epilogue_bias_activation_residual(ops, act):
    rng = np.random.default_rng(3)
    M, N, K, G = 5, 384, 512, 128
    x, q, s, z = make_case(rng, M, N, K, G, 4, "bf16")
    bias = bf16_round(rng.normal(0, 0.5, N).astype(np.float32))
    res = bf16_round(rng.normal(0, 1, (M, N)).astype(np.float32))
    pre = gemm_ref.gemm_a16wx(x, q, s, z, G, 4, alpha=1.5, bias=bias, act=act, ft="bf16")
    ref = bf16_round(pre + res)
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, 4)
    y = ops.gemm_lowp(to_dev(x, "bf16"), pw, bias=to_dev(bias, "bf16"), residual=to_dev(res, "bf16"), act=act, alpha=1.5)
    assert_close(y.float().cpu().numpy(), ref, "bf16", what=f"act={act}", pre=pre)


def test_workspace_counters_path_without_sync_buffer(ops):
    """sync == NULL: counters live in the workspace and are cleared per call (memset node)."""
    rng = np.random.default_rng(4)
    M, N, K, G = 1, 512, 4096, 128
    x, q, s, z = make_case(rng, M, N, K, G, 4, "bf16")
    ref = gemm_ref.gemm_a16wx(x, q, s, z, G, 4, ft="bf16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, 4)
    sc = ops.Scratch(ops.lowp_workspace_bytes(4, M, N, K, G))
    sc.ws.fill_(0xFF)  # poisoned scratch must not matter
    y1 = ops.gemm_lowp(to_dev(x, "bf16"), pw, scratch=sc, use_sync=False)
    y2 = ops.gemm_lowp(to_dev(x, "bf16"), pw, scratch=sc, use_sync=False)
    assert_close(y1.float().cpu().numpy(), ref, "bf16")
    assert torch.equal(y1, y2)


def test_error_behaviour(ops):
    from dash_infer_amd import capi
    rng = np.random.default_rng(5)
    x, q, s, z = make_case(rng, 1, 64, 128, 64, 8, "bf16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), 64, 8)
    with pytest.raises(capi.DihipError) as e:  # fp32 activations are not an A16 type (gemm_a16w8.cpp)
        ops.gemm_lowp(to_dev(x), pw)
    assert e.value.code == capi.PARAM_ERROR
    with pytest.raises(capi.DihipError) as e:
        ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), 48, 8)  # GroupSize % 32
    assert e.value.code == capi.PARAM_ERROR
    empty = ops.gemm_lowp(torch.empty(0, 128, dtype=torch.bfloat16, device="cuda"), pw)  # empty batch
    assert empty.shape == (0, 64)


# ------------------------------------------------------------------ BASELINE shapes ----------
QWEN7B = {"qkv": (3584, 4608), "o": (3584, 3584), "gate": (3584, 18944), "down": (18944, 3584)}


@pytest.mark.parametrize("layer", list(QWEN7B))
@pytest.mark.parametrize("wbits,G", [(8, -1), (4, 128)])
def test_qwen2_7b_layer_shapes_m1_vs_c_oracle(ops, layer, wbits, G):
    """configs[1] (int8 per-channel) and configs[2] (int4 g128) linear layers at batch 1, checked
    directly against the plain-C oracle (sequential f32 CPU_SubC_Ref semantics)."""
    K, N = QWEN7B[layer]
    rng = np.random.default_rng(N + wbits)
    x, q, s, z = make_case(rng, 1, N, K, G, wbits, "bf16")
    ref = cbind.gemm_a16wx(x, q, s, z, G, wbits, ft="bf16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
    y = ops.gemm_lowp(to_dev(x, "bf16"), pw)
    out = y.float().cpu().numpy()
    # sequential-f32 oracle vs f32-MFMA kernel: both round once to bf16; allow 1 ulp + eps
    assert_close(out, ref, "bf16", what=f"{layer} w{wbits}")


@pytest.mark.parametrize("wbits,G", [(4, 128), (8, -1)])
def test_batch_invariance_and_determinism_full_size(ops, wbits, G):
    """Size-independent properties at BASELINE size (gate proj 3584 -> 18944, batch 32):
    within one kernel family (decode GEMV: M <= 4, activations resident in LDS; small-batch kernel:
    4 < M <= 32) row m of a batched
    call is bit-identical to a smaller-batch call on that row; across the two families rows agree to
    one FT ulp; repeated launches are bit-identical (deterministic reductions), and the op is linear
    in x within FT rounding."""
    K, N = QWEN7B["gate"]
    rng = np.random.default_rng(9)
    x, q, s, z = make_case(rng, 32, N, K, G, wbits, "bf16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
    xd = to_dev(x, "bf16")
    y32 = ops.gemm_lowp(xd, pw)
    y32b = ops.gemm_lowp(xd, pw)
    assert torch.equal(y32, y32b)
    assert ops.gemv_plan(wbits, 4, N, K, G) is not None and ops.gemv_plan(wbits, 8, N, K, G) is None
    y4 = ops.gemm_lowp(xd[:4].contiguous(), pw)            # decode GEMV family (activations resident in LDS)
    assert torch.equal(y4, ops.gemm_lowp(xd[:4].contiguous(), pw))
    for m in (0, 3):
        y1 = ops.gemm_lowp(xd[m:m + 1].contiguous(), pw)
        assert torch.equal(y1[0], y4[m]), f"row {m} differs between M=1 and M=4"
    y24 = ops.gemm_lowp(xd[:24].contiguous(), pw)          # small-batch family
    assert torch.equal(y24, y32[:24])
    y8 = ops.gemm_lowp(xd[:8].contiguous(), pw)
    assert torch.equal(y8, y32[:8])
    assert_close(y4.float().cpu().numpy(), y32[:4].float().cpu().numpy(), "bf16", what="GEMV vs small-batch family")
    # linearity: f(2x) == 2 f(x) exactly (power-of-two scaling commutes with every rounding)
    y2 = ops.gemm_lowp((xd[:4] * 2).contiguous(), pw)
    assert torch.equal(y2, y4 * 2)
    # spot-check 64 random columns of 2 rows against the f64 oracle
    cols = rng.choice(N, 64, replace=False)
    w = gemm_ref.dequant(q, s, z, G, wbits)[:, cols].astype(np.float64)
    ref = bf16_round((x[[0, 31]].astype(np.float64) @ w).astype(np.float32))
    assert_close(y32[[0, 31]][:, cols].float().cpu().numpy(), ref, "bf16", what="spot check")


# ------------------------------------------------------------------ fused decode-step forms --
@pytest.mark.parametrize("M", [1, 3, 8, 32])
@pytest.mark.parametrize("wbits,G", [(4, 128), (8, -1)])
def test_fused_norm_gemm_swiglu_addto(ops, M, wbits, G):
    rng = np.random.default_rng(M + wbits)
    K, I = 512, 1152
    eps = 1e-6
    h = rng.normal(0, 1.5, (M, K)).astype(np.float32)
    gamma = bf16_round(rng.normal(1, 0.1, K).astype(np.float32))
    xn = bf16_round(glue.rmsnorm(h, gamma, eps))
    sc = ops.Scratch(max(ops.lowp_workspace_bytes(wbits, M, I, K, G), ops.lowp_workspace_bytes(wbits, M, K, I, G)))
    hd, gd = to_dev(h), to_dev(gamma, "bf16")

    # qkv-like: norm + gemm + bias
    _, q1, s1, z1 = make_case(rng, 1, 768, K, G, wbits, "bf16")
    bias = bf16_round(rng.normal(0, 0.3, 768).astype(np.float32))
    p1 = ops.pack_lowp(to_dev(q1), to_dev(s1, "bf16"), to_dev(z1, "bf16"), G, wbits)
    y = ops.fused_norm_gemm(hd, gd, eps, p1, to_dev(bias, "bf16"), sc)
    ref = gemm_ref.gemm_a16wx(xn, q1, s1, z1, G, wbits, bias=bias, ft="bf16")
    assert_close(y.float().cpu().numpy(), ref, "bf16", what="norm_gemm")

    # gate/up SwiGLU
    _, qg, sg, zg = make_case(rng, 1, I, K, G, wbits, "bf16")
    _, qu, su, zu = make_case(rng, 1, I, K, G, wbits, "bf16")
    pg = ops.pack_lowp(to_dev(qg), to_dev(sg, "bf16"), to_dev(zg, "bf16"), G, wbits)
    pu = ops.pack_lowp(to_dev(qu), to_dev(su, "bf16"), to_dev(zu, "bf16"), G, wbits)
    act = ops.fused_norm_swiglu(hd, gd, eps, pg, pu, sc)
    g = gemm_ref.gemm_a16wx(xn, qg, sg, zg, G, wbits, round_out=False).astype(np.float64)
    u = gemm_ref.gemm_a16wx(xn, qu, su, zu, G, wbits, round_out=False).astype(np.float64)
    ref = bf16_round(((g / (1 + np.exp(-g))) * u).astype(np.float32))
    assert_close(act.float().cpu().numpy(), ref, "bf16", what="swiglu")

    # down proj + residual into the f32 hidden stream
    _, qd, sd, zd = make_case(rng, 1, K, I, G, wbits, "bf16")
    pd = ops.pack_lowp(to_dev(qd), to_dev(sd, "bf16"), to_dev(zd, "bf16"), G, wbits)
    a_host = act.float().cpu().numpy()
    h2 = ops.fused_gemm_addto(act, pd, hd, sc)
    ref = h + gemm_ref.gemm_a16wx(a_host, qd, sd, zd, G, wbits, round_out=False)
    np.testing.assert_allclose(h2.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    torch.cuda.synchronize()
    assert int(sc.sync.sum()) == 0  # arrival counters are left zero


def test_lm_head_and_argmax(ops):
    rng = np.random.default_rng(11)
    K, V = 896, 5000 + 37
    eps = 1e-6
    W = bf16_round(rng.normal(0, 0.05, (K, V)).astype(np.float32))
    gamma = bf16_round(rng.normal(1, 0.1, K).astype(np.float32))
    pw = ops.pack_dense(to_dev(W, "bf16"))
    sc = ops.Scratch(int(ops.lib().dihip_dense_workspace_bytes(32, V, K)))
    for M in (1, 4, 9):
        h = rng.normal(0, 1, (M, K)).astype(np.float32)
        xn = bf16_round(glue.rmsnorm(h, gamma, eps))
        ref = xn.astype(np.float64) @ W.astype(np.float64)
        logits = ops.lm_head(to_dev(h), to_dev(gamma, "bf16"), eps, pw, sc)
        np.testing.assert_allclose(logits.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
        ids = ops.argmax(logits)
        np.testing.assert_array_equal(ids.cpu().numpy(), np.argmax(logits.cpu().numpy(), axis=-1))
    # ties resolve to the lowest index
    t = torch.zeros(2, 1000, device="cuda")
    t[0, 400] = t[0, 7] = 5.0
    t[1, 999] = 1.0
    assert ops.argmax(t).tolist() == [7, 999]


# ------------------------------------------------------------------ batched decode shapes ----
CFG4_RANK = {"qkv": (8192, 1280), "o": (1024, 8192), "gate": (8192, 3712), "down": (3712, 8192)}  # Qwen2-72B, TP = 8


@pytest.mark.parametrize("layer", list(CFG4_RANK))
def test_config4_rank_shapes_m16_vs_c_oracle(ops, layer):
    """BASELINE configs[3]: Qwen2-72B int4 g128, TP = 8, batch 16 -- per-rank linear layers (the activations
    do not fit in LDS: small-batch register-resident kernel)."""
    K, N = CFG4_RANK[layer]
    rng = np.random.default_rng(N)
    x, q, s, z = make_case(rng, 16, N, K, 128, 4, "bf16")
    ref = cbind.gemm_a16wx(x, q, s, z, 128, 4, ft="bf16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), 128, 4)
    y = ops.gemm_lowp(to_dev(x, "bf16"), pw)
    assert_close(y.float().cpu().numpy(), ref, "bf16", what=f"cfg4 {layer}")


@pytest.mark.parametrize("wbits,G", [(4, 128), (8, -1), (8, 128), (4, 256)])
@pytest.mark.parametrize("M", [2, 17, 32])
def test_small_batch_kernel_all_epilogues(ops, wbits, G, M):
    """configs[2] regime (batch 32) and odd batch sizes through every fused form at a K that does not fit LDS."""
    rng = np.random.default_rng(M * 10 + wbits)
    K, N = 4096, 272
    x, q, s, z = make_case(rng, M, N, K, G, wbits, "bf16")
    _, q2, s2, z2 = make_case(rng, 1, N, K, G, wbits, "bf16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
    pw2 = ops.pack_lowp(to_dev(q2), to_dev(s2, "bf16"), to_dev(z2, "bf16"), G, wbits)
    bias = bf16_round(rng.normal(0, 0.5, N).astype(np.float32))
    xd = to_dev(x, "bf16")
    y = ops.gemm_lowp(xd, pw, bias=to_dev(bias, "bf16"), act="silu", alpha=0.5)
    ref = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, alpha=0.5, bias=bias, act="silu", ft="bf16")
    assert_close(y.float().cpu().numpy(), ref, "bf16", what="std")
    sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
    h = rng.normal(0, 1, (M, N)).astype(np.float32)
    hd = torch.from_numpy(h).cuda()
    out = ops.fused_gemm_addto(xd, pw, hd, sc)
    ref2 = h + gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, ft="f32")
    np.testing.assert_allclose(out.cpu().numpy(), ref2, rtol=2e-3, atol=2e-3 * np.abs(ref2).max())
    # SwiGLU over a (gate, up) pair from the f32 hidden stream (M > 4: norm runs as its own launch)
    hk = rng.normal(0, 1.5, (M, K)).astype(np.float32)
    gamma = bf16_round(rng.normal(1, 0.1, K).astype(np.float32))
    act = ops.fused_norm_swiglu(torch.from_numpy(hk).cuda(), to_dev(gamma, "bf16"), 1e-6, pw, pw2, sc)
    xn = bf16_round(glue.rmsnorm(hk, gamma, 1e-6))
    g_ = gemm_ref.gemm_a16wx(xn, q, s, z, G, wbits, ft="f32")
    u_ = gemm_ref.gemm_a16wx(xn, q2, s2, z2, G, wbits, ft="f32")
    ref3 = bf16_round(glue.silu(g_) * u_)
    assert_close(act.float().cpu().numpy(), ref3, "bf16", what="swiglu", pre=ref3)


@pytest.mark.gpu
@pytest.mark.parametrize("wbits,G", [(4, 128), (8, -1)])
@pytest.mark.parametrize("M", [7, 16, 17, 32])
def test_frag32_activation_layout_chain(ops, wbits, G, M):
    """FRAG32 (MFMA-fragment-major) activations between SwiGLU and the down projection: converters round-trip,
    and the fused chain is bit-identical to the row-major chain (same kernels, same arithmetic, other addresses)."""
    rng = np.random.default_rng(M * 7 + wbits)
    K, N = 4096, 272          # K too large for the LDS-resident (batch <= 16) kernel: the small-batch kernel runs
    K2, N2 = 4096, 304        # second projection: consumes the [M, K2] activations
    x, q, s, z = make_case(rng, M, N, K, G, wbits, "bf16")
    xd = to_dev(x, "bf16")
    # converters
    xf = ops.act_to_frag(xd)
    assert xf.numel() == (32 if M > 16 else 16) * K
    back = ops.act_from_frag(xf, M, K)
    assert torch.equal(back.view(torch.int16), xd.view(torch.int16))
    mt = 2 if M > 16 else 1
    m_i, k_i = M - 1, 1234
    idx = ((((k_i // 32) * mt + m_i // 16) * 64 + ((k_i % 32) // 8) * 16 + m_i % 16) * 8 + k_i % 8)
    assert xf.view(torch.int16)[idx].item() == xd.view(torch.int16)[m_i, k_i].item()   # layout formula of the header
    # gate/up -> down, row-major vs FRAG32
    _, qg, sg, zg = make_case(rng, 1, K2, K, G, wbits, "bf16")
    _, qu, su, zu = make_case(rng, 1, K2, K, G, wbits, "bf16")
    _, qd, sd, zd = make_case(rng, 1, N2, K2, G, wbits, "bf16")
    pg = ops.pack_lowp(to_dev(qg), to_dev(sg, "bf16"), to_dev(zg, "bf16"), G, wbits)
    pu = ops.pack_lowp(to_dev(qu), to_dev(su, "bf16"), to_dev(zu, "bf16"), G, wbits)
    pd = ops.pack_lowp(to_dev(qd), to_dev(sd, "bf16"), to_dev(zd, "bf16"), G, wbits)
    frag_ok = ops.prefers_frag(pg, M, dual=True) and ops.prefers_frag(pd, M)
    assert frag_ok                   # every 4 < M <= 32 is served by the small-batch kernel
    assert not ops.prefers_frag(pd, 1) and not ops.prefers_frag(pd, 33)
    sc = ops.Scratch(max(ops.lowp_workspace_bytes(wbits, M, K2, K, G), ops.lowp_workspace_bytes(wbits, M, N2, K2, G)))
    hk = torch.from_numpy(rng.normal(0, 1.5, (M, K)).astype(np.float32)).cuda()
    gamma = to_dev(bf16_round(rng.normal(1, 0.1, K).astype(np.float32)), "bf16")
    h0 = torch.from_numpy(rng.normal(0, 1, (M, N2)).astype(np.float32)).cuda()
    act_rm = ops.fused_norm_swiglu(hk, gamma, 1e-6, pg, pu, sc)
    out_rm = ops.fused_gemm_addto(act_rm, pd, h0, sc)
    if not frag_ok:
        # an explicit FRAG32 request forces the small-batch kernel; the row-major chain ran on the LDS-resident
        # kernel (another summation order), so the two agree to rounding only
        act_fr = ops.fused_norm_swiglu(hk, gamma, 1e-6, pg, pu, sc, y_layout=ops.ACT_FRAG32)
        a_rm, a_fr = act_rm.float().cpu().numpy(), ops.act_from_frag(act_fr, M, K2).float().cpu().numpy()
        np.testing.assert_allclose(a_fr, a_rm, rtol=2e-2, atol=2e-2 * np.abs(a_rm).max())
        out_fr = ops.fused_gemm_addto(act_fr, pd, h0, sc, x_layout=ops.ACT_FRAG32, M=M)
        np.testing.assert_allclose(out_fr.cpu().numpy(), out_rm.cpu().numpy(), rtol=2e-2, atol=2e-2 * float(out_rm.abs().max()))
        with pytest.raises(Exception):   # M <= 4 runs on the LDS-resident kernel only: refused loudly
            ops.fused_gemm_addto(act_fr, pd, h0[:1], sc, x_layout=ops.ACT_FRAG32, M=1)
        return
    act_fr = ops.fused_norm_swiglu(hk, gamma, 1e-6, pg, pu, sc, y_layout=ops.ACT_FRAG32)
    assert torch.equal(ops.act_from_frag(act_fr, M, K2).view(torch.int16), act_rm.view(torch.int16))
    out_fr = ops.fused_gemm_addto(act_fr, pd, h0, sc, x_layout=ops.ACT_FRAG32, M=M)
    assert torch.equal(out_fr, out_rm)
    with pytest.raises(Exception):   # M <= 4 runs on the LDS-resident kernel only: the layout is refused loudly
        ops.fused_gemm_addto(act_fr, pd, h0[:1], sc, x_layout=ops.ACT_FRAG32, M=1)


@pytest.mark.gpu
@pytest.mark.parametrize("wbits,G", [(4, 128), (8, -1), (8, 128)])
@pytest.mark.parametrize("M", [17, 32])
def test_panel_kernel_full_size_mlp(ops, wbits, G, M):
    """BASELINE configs[2] MLP at full size through the FRAG32 / panel path (gate/up: one K-slice, 148 panels;
    down: split-K over 9 slices + reduce) against the row-major whole-column kernels and the f64 oracle."""
    rng = np.random.default_rng(M + wbits)
    K, I = QWEN7B["gate"]                # 3584 -> 18944
    hk = torch.from_numpy(rng.normal(0, 1.0, (M, K)).astype(np.float32)).cuda()
    gamma = to_dev(bf16_round(rng.normal(1, 0.1, K).astype(np.float32)), "bf16")
    _, qg, sg, zg = make_case(rng, 1, I, K, G, wbits, "bf16")
    _, qu, su, zu = make_case(rng, 1, I, K, G, wbits, "bf16")
    _, qd, sd, zd = make_case(rng, 1, K, I, G, wbits, "bf16")
    pg = ops.pack_lowp(to_dev(qg), to_dev(sg, "bf16"), to_dev(zg, "bf16"), G, wbits)
    pu = ops.pack_lowp(to_dev(qu), to_dev(su, "bf16"), to_dev(zu, "bf16"), G, wbits)
    pd = ops.pack_lowp(to_dev(qd), to_dev(sd, "bf16"), to_dev(zd, "bf16"), G, wbits)
    sc = ops.Scratch(max(ops.lowp_workspace_bytes(wbits, M, I, K, G), ops.lowp_workspace_bytes(wbits, M, K, I, G)))
    h0 = torch.from_numpy(rng.normal(0, 1, (M, K)).astype(np.float32)).cuda()
    act_rm = ops.fused_norm_swiglu(hk, gamma, 1e-6, pg, pu, sc)                                   # row-major out
    act_fr = ops.fused_norm_swiglu(hk, gamma, 1e-6, pg, pu, sc, y_layout=ops.ACT_FRAG32)
    a_rm = act_rm.float().cpu().numpy()
    a_fr = ops.act_from_frag(act_fr, M, I).float().cpu().numpy()
    assert_close(a_fr, a_rm, "bf16", what="swiglu FRAG32 vs row-major output", pre=a_rm)
    out_rm = ops.fused_gemm_addto(act_rm, pd, h0, sc)                                              # whole-column kernel
    out_fr = ops.fused_gemm_addto(ops.act_to_frag(act_rm), pd, h0, sc, x_layout=ops.ACT_FRAG32, M=M)  # panel, split-K
    o_rm, o_fr = out_rm.cpu().numpy(), out_fr.cpu().numpy()
    np.testing.assert_allclose(o_fr, o_rm, rtol=2e-3, atol=2e-3 * np.abs(o_rm - h0.cpu().numpy()).max())
    assert torch.equal(out_fr, ops.fused_gemm_addto(ops.act_to_frag(act_rm), pd, h0, sc, x_layout=ops.ACT_FRAG32, M=M))
    # f64 oracle on 48 random columns of the down projection (input = the kernel's own bf16 activations)
    cols = rng.choice(K, 48, replace=False)
    wd = gemm_ref.dequant(qd, sd, zd, G, wbits)[:, cols].astype(np.float64)
    ref = h0.cpu().numpy()[:, cols] + (a_rm.astype(np.float64) @ wd)
    np.testing.assert_allclose(o_fr[:, cols], ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("wbits,G,M", [(4, 128, 5), (4, 128, 16), (4, 128, 32), (4, 256, 16), (8, -1, 16), (8, -1, 32), (8, 128, 9)])
def test_kslice_kernel_full_size_mlp(ops, wbits, G, M):
    """Register-resident K-slice kernel (gemm_kslice_kernel.hpp) on the BASELINE configs[2] MLP at full size:
    SwiGLU epilogue with one workgroup-level K-slice (gate/up, 7 waves), residual epilogue through the split-K slab
    (down: 37 slices = 5 workgroup slices), plain epilogue with bias.  DIHIP_GEMM_KSLICE=2 forces it for every
    eligible shape (by default it serves weight matrices / SwiGLU pairs of >= 24 MB); checked against the panel kernel
    (DIHIP_GEMM_KSLICE=0; other summation order over K) and the f64 oracle."""
    import os
    rng = np.random.default_rng(3 * M + wbits + G)
    K, I = QWEN7B["gate"]                # 3584 -> 18944
    hk = torch.from_numpy(rng.normal(0, 1.0, (M, K)).astype(np.float32)).cuda()
    gamma = to_dev(bf16_round(rng.normal(1, 0.1, K).astype(np.float32)), "bf16")
    _, qg, sg, zg = make_case(rng, 1, I, K, G, wbits, "bf16")
    _, qu, su, zu = make_case(rng, 1, I, K, G, wbits, "bf16")
    _, qd, sd, zd = make_case(rng, 1, K, I, G, wbits, "bf16")
    pg = ops.pack_lowp(to_dev(qg), to_dev(sg, "bf16"), to_dev(zg, "bf16"), G, wbits)
    pu = ops.pack_lowp(to_dev(qu), to_dev(su, "bf16"), to_dev(zu, "bf16"), G, wbits)
    pd = ops.pack_lowp(to_dev(qd), to_dev(sd, "bf16"), to_dev(zd, "bf16"), G, wbits)
    sc = ops.Scratch(max(ops.lowp_workspace_bytes(wbits, M, I, K, G), ops.lowp_workspace_bytes(wbits, M, K, I, G)))
    h0 = torch.from_numpy(rng.normal(0, 1, (M, K)).astype(np.float32)).cuda()
    NQ = 4608
    _, qq, sq, zq = make_case(rng, 1, NQ, K, G, wbits, "bf16")
    pq = ops.pack_lowp(to_dev(qq), to_dev(sq, "bf16"), to_dev(zq, "bf16"), G, wbits)
    bias = to_dev(bf16_round(rng.normal(0, 0.5, NQ).astype(np.float32)), "bf16")
    old = os.environ.get("DIHIP_GEMM_KSLICE")

    def run(mode):
        os.environ["DIHIP_GEMM_KSLICE"] = mode
        act = ops.fused_norm_swiglu(hk, gamma, 1e-6, pg, pu, sc, y_layout=ops.ACT_FRAG32)
        out = ops.fused_gemm_addto(act, pd, h0, sc, x_layout=ops.ACT_FRAG32, M=M)
        y = ops.fused_norm_gemm(hk, gamma, 1e-6, pq, bias, sc)
        return ops.act_from_frag(act, M, I), out, y

    try:
        a_pan, o_pan, y_pan = run("0")
        a_ksl, o_ksl, y_ksl = run("2")
        a_ks2, o_ks2, y_ks2 = run("2")
    finally:
        if old is None:
            os.environ.pop("DIHIP_GEMM_KSLICE", None)
        else:
            os.environ["DIHIP_GEMM_KSLICE"] = old
    assert torch.equal(a_ksl.view(torch.int16), a_ks2.view(torch.int16)) and torch.equal(o_ksl, o_ks2)   # deterministic
    assert torch.equal(y_ksl.view(torch.int16), y_ks2.view(torch.int16))
    yp, yk = y_pan.float().cpu().numpy(), y_ksl.float().cpu().numpy()
    assert_close(yk, yp, "bf16", what="qkv (bias epilogue): K-slice vs panel kernel", pre=yp)
    ap, ak = a_pan.float().cpu().numpy(), a_ksl.float().cpu().numpy()
    assert_close(ak, ap, "bf16", what="swiglu: K-slice vs panel kernel", pre=ap)
    op_, ok_ = o_pan.cpu().numpy(), o_ksl.cpu().numpy()
    h0n = h0.cpu().numpy()
    # the two down projections start from activations that may differ in the last bf16 bit
    np.testing.assert_allclose(ok_, op_, rtol=1e-2, atol=1e-2 * np.abs(op_ - h0n).max())
    # f64 oracle: 48 random columns of both projections (down: input = the kernel's own bf16 activations)
    cols = rng.choice(I, 48, replace=False)
    xn = bf16_round(glue.rmsnorm(hk.cpu().numpy(), gamma.float().cpu().numpy(), 1e-6)).astype(np.float64)
    g_ = xn @ gemm_ref.dequant(qg, sg, zg, G, wbits)[:, cols].astype(np.float64)
    u_ = xn @ gemm_ref.dequant(qu, su, zu, G, wbits)[:, cols].astype(np.float64)
    ref_a = glue.silu(g_) * u_
    np.testing.assert_allclose(ak[:, cols], ref_a, rtol=1e-2, atol=1e-2 * np.abs(ref_a).max())
    cols = rng.choice(K, 48, replace=False)
    ref_o = h0n[:, cols] + ak.astype(np.float64) @ gemm_ref.dequant(qd, sd, zd, G, wbits)[:, cols].astype(np.float64)
    np.testing.assert_allclose(ok_[:, cols], ref_o, rtol=2e-3, atol=2e-3 * np.abs(ref_o).max())


# Qwen2-7B under TP = 8 (tp.py: every KV head on two ranks, its 7 query heads split 4 + 3; 148 FFN groups split 19 / 18):
# the per-rank linear layers of the ranks with the larger and the smaller share
TP8_7B = [("qkv_r0", 3584, 768), ("qkv_r1", 3584, 640), ("o_r0", 512, 3584), ("o_r1", 384, 3584),
          ("gate_r0", 3584, 2432), ("gate_r7", 3584, 2304), ("down_r0", 2432, 3584), ("down_r7", 2304, 3584)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,K,N", TP8_7B)
@pytest.mark.parametrize("M", [1, 32])
def test_qwen7b_tp8_rank_shapes(ops, name, K, N, M):
    """The scale bench runs the decode step at TP = 2 / 4 / 8: every per-rank shape must be served (int4 g128 and
    int8 per-channel, batch 1 and 32, fused forms as the decoder calls them) and match the oracle."""
    rng = np.random.default_rng(K + N + M)
    for wbits, G in ((4, 128), (8, -1)):
        x, q, s, z = make_case(rng, M, N, K, G, wbits, "bf16")
        pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
        sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
        if name.startswith(("o_", "down_")):        # residual form from bf16 activations
            h = rng.normal(0, 1, (M, N)).astype(np.float32)
            out = ops.fused_gemm_addto(to_dev(x, "bf16"), pw, torch.from_numpy(h).cuda(), sc)
            ref = h + gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, ft="f32")
            np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
            continue
        hk = rng.normal(0, 1.5, (M, K)).astype(np.float32)
        gamma = bf16_round(rng.normal(1, 0.1, K).astype(np.float32))
        xn = bf16_round(glue.rmsnorm(hk, gamma, 1e-6))
        hd, gd = torch.from_numpy(hk).cuda(), to_dev(gamma, "bf16")
        if name.startswith("qkv"):
            bias = bf16_round(rng.normal(0, 0.5, N).astype(np.float32))
            y = ops.fused_norm_gemm(hd, gd, 1e-6, pw, to_dev(bias, "bf16"), sc)
            ref = gemm_ref.gemm_a16wx(xn, q, s, z, G, wbits, bias=bias, ft="bf16")
            assert_close(y.float().cpu().numpy(), ref, "bf16", what=f"{name} W{wbits}")
        else:
            _, q2, s2, z2 = make_case(rng, 1, N, K, G, wbits, "bf16")
            pw2 = ops.pack_lowp(to_dev(q2), to_dev(s2, "bf16"), to_dev(z2, "bf16"), G, wbits)
            act = ops.fused_norm_swiglu(hd, gd, 1e-6, pw, pw2, sc)
            g_ = gemm_ref.gemm_a16wx(xn, q, s, z, G, wbits, ft="f32")
            u_ = gemm_ref.gemm_a16wx(xn, q2, s2, z2, G, wbits, ft="f32")
            ref = bf16_round(glue.silu(g_) * u_)
            assert_close(act.float().cpu().numpy(), ref, "bf16", what=f"{name} W{wbits} swiglu", pre=ref)


@pytest.mark.gpu
@pytest.mark.parametrize("wbits,G,M", [(4, 128, 32), (4, 128, 8), (8, -1, 17)])
def test_residual_gemm_with_fused_norm(ops, wbits, G, M):
    """dihip_fused_gemm_addto_norm / dihip_prenorm_gemm / dihip_prenorm_swiglu (batched decode, one launch pair less per
    residual GEMM): h_out must be bit-identical to dihip_fused_gemm_addto, and the GEMMs fed with the normalised rows
    bit-identical to the forms that normalise h_out themselves -- on a split-K plan (down projection: the norm rides on the
    slab reduction) and on a plan without one (wide N: separate norm launch)."""
    rng = np.random.default_rng(5 * M + wbits)
    hidden, inter = 3584, 18944
    for K, N in ((inter, hidden), (hidden, 8192)):     # split-K slab / no slab
        x, q, s, z = make_case(rng, M, N, K, G, wbits, "bf16")
        pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
        _, q1, s1, z1 = make_case(rng, 1, 1024, N, G, wbits, "bf16")        # consumers of the normalised rows
        _, q2, s2, z2 = make_case(rng, 1, 1024, N, G, wbits, "bf16")
        p1 = ops.pack_lowp(to_dev(q1), to_dev(s1, "bf16"), to_dev(z1, "bf16"), G, wbits)
        p2 = ops.pack_lowp(to_dev(q2), to_dev(s2, "bf16"), to_dev(z2, "bf16"), G, wbits)
        sc = ops.Scratch(max(ops.lowp_workspace_bytes(wbits, M, N, K, G), ops.lowp_workspace_bytes(wbits, M, 1024, N, G)))
        h = torch.from_numpy(rng.normal(0, 1, (M, N)).astype(np.float32)).cuda()
        gamma = to_dev(bf16_round(rng.normal(1, 0.1, N).astype(np.float32)), "bf16")
        bias = to_dev(bf16_round(rng.normal(0, 0.5, 1024).astype(np.float32)), "bf16")
        xd = to_dev(x, "bf16")
        frag = ops.prefers_frag(pw, M)
        xin = ops.act_to_frag(xd) if frag else xd
        lay = ops.ACT_FRAG32 if frag else ops.ACT_ROWMAJOR
        ref_h = ops.fused_gemm_addto(xin, pw, h, sc, x_layout=lay, M=M)
        for consumer_dual in (False, True):
            cfrag = ops.prefers_frag(p1, M, dual=consumer_dual)
            clay = ops.ACT_FRAG32 if cfrag else ops.ACT_ROWMAJOR
            xn = torch.zeros(ops.act_frag_numel(M, N) if cfrag else M * N, dtype=torch.bfloat16, device="cuda")
            out_h = ops.fused_gemm_addto_norm(xin, pw, h, sc, gamma, 1e-6, xn, x_layout=lay, xnorm_layout=clay, M=M)
            assert torch.equal(out_h, ref_h)
            xn_rm = ops.act_from_frag(xn, M, N) if cfrag else xn.view(M, N)
            ref_n = bf16_round(glue.rmsnorm(ref_h.cpu().numpy(), gamma.float().cpu().numpy(), 1e-6))
            assert_close(xn_rm.float().cpu().numpy(), ref_n, "bf16", what="fused norm rows", pre=ref_n)
            if consumer_dual:
                y0 = ops.fused_norm_swiglu(ref_h, gamma, 1e-6, p1, p2, sc)
                y1 = ops.prenorm_swiglu(xn, p1, p2, sc, M, x_layout=clay)
            else:
                y0 = ops.fused_norm_gemm(ref_h, gamma, 1e-6, p1, bias, sc)
                y1 = ops.prenorm_gemm(xn, p1, bias, sc, M, x_layout=clay)
            assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))


# ------------------------------------------------------------------------------------------------------------------
# Context-phase GEMM (gemm_prefill_kernel.hpp: M >= 64 rows, 128 x 256 workgroup tiles, A through LDS): every epilogue, ragged M
# (rows past M clamped on load, masked on store), ragged N (last column tile / last workgroup partly outside), groups of one
# k-tile (W4 g128, W8 g64), wider groups (W4 g256, W8 g128) and per-channel, against the oracle.
@pytest.mark.gpu
@pytest.mark.parametrize("wbits,G", [(4, 128), (4, 256), (4, -1), (8, -1), (8, 64), (8, 128)])
@pytest.mark.parametrize("M,N,K", [(64, 512, 512), (200, 261, 1024), (999, 1056, 768), (2048, 640, 3584)])
def test_prefill_gemm_all_epilogues(ops, M, N, K, wbits, G):
    rng = np.random.default_rng(M + N + K + wbits + abs(G))
    x, q, s, z = make_case(rng, M, N, K, G, wbits, "bf16", style="iq")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
    xd = to_dev(x, "bf16")
    sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
    # STD: alpha, bias, activation, FT residual
    bias = bf16_round(rng.normal(0, 0.3, N).astype(np.float32))
    res = bf16_round(rng.normal(0, 1, (M, N)).astype(np.float32))
    y = ops.gemm_lowp(xd, pw, bias=to_dev(bias, "bf16"), residual=to_dev(res, "bf16"), act="silu", alpha=0.5, scratch=sc)
    pre = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, alpha=0.5, bias=bias, act="silu", ft="bf16")
    assert_close(y.float().cpu().numpy(), bf16_round(pre + res), "bf16", what=f"prefill STD M{M} N{N} K{K}", pre=pre)
    # f32 hidden-stream update
    h = rng.normal(0, 1.5, (M, N)).astype(np.float32)
    h2 = ops.fused_gemm_addto(xd, pw, to_dev(h), sc, M=M)
    ref = h + gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, round_out=False)
    np.testing.assert_allclose(h2.cpu().numpy(), ref, rtol=2e-5, atol=3e-5 * np.abs(ref).max())
    # SwiGLU over a gate / up pair (prenorm entry: PRO_PLAIN on normalised rows)
    _, qu, su, zu = make_case(rng, 1, N, K, G, wbits, "bf16", style="iq")
    pu = ops.pack_lowp(to_dev(qu), to_dev(su, "bf16"), to_dev(zu, "bf16"), G, wbits)
    act = ops.prenorm_swiglu(xd, pw, pu, sc, M)
    g = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, round_out=False).astype(np.float64)
    u = gemm_ref.gemm_a16wx(x, qu, su, zu, G, wbits, round_out=False).astype(np.float64)
    assert_close(act.float().cpu().numpy(), bf16_round(((g / (1 + np.exp(-g))) * u).astype(np.float32)), "bf16", what="prefill SwiGLU")


# ------------------------------------------------------------------------------------------------------------------
# Context-phase GEMM, tail split (gemm_prefill_kernel.hpp, PrefillArgs::tail_cb): shapes whose grid of 128 x 256 tiles ends in a
# round that fills at most half of the chip -- the column blocks of that round are split in K and added by a reduction launch.
# Every epilogue, groups of one k-tile / two k-tiles / per-channel, ragged M and N, against the oracle; the split must be ON for
# these shapes on this GPU (dihip_gemm_prefill_tail_parts), or the test says nothing.
@pytest.mark.gpu
@pytest.mark.parametrize("wbits,G,M,N,K", [(4, 128, 2048, 4608, 1024),     # the qkv projection's grid: 288 tiles -> 256 + 32 x 8 parts
                                           (4, 256, 2000, 4600, 1024),     # ragged rows / columns, a group = two k-tiles
                                           (8, -1, 2048, 4608, 512),       # per-channel: a part closes its sum with the one scale
                                           (4, 128, 1100, 7400, 768)])     # 9 row blocks x 29 column blocks = 261 tiles: a 5-tile tail
def test_prefill_gemm_tail_split(ops, wbits, G, M, N, K):
    from dash_infer_amd.capi import lib
    rng = np.random.default_rng(M + N + K + wbits)
    x, q, s, z = make_case(rng, M, N, K, G, wbits, "bf16", style="iq")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
    xd = to_dev(x, "bf16")
    sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
    parts = int(lib().dihip_gemm_prefill_tail_parts(wbits, M, N, K, G, 0))
    assert parts > 1, "this shape must run with a split tail on a 256-CU part"
    # STD: alpha, bias, activation, FT residual
    bias = bf16_round(rng.normal(0, 0.3, N).astype(np.float32))
    res = bf16_round(rng.normal(0, 1, (M, N)).astype(np.float32))
    y = ops.gemm_lowp(xd, pw, bias=to_dev(bias, "bf16"), residual=to_dev(res, "bf16"), act="silu", alpha=0.5, scratch=sc)
    pre = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, alpha=0.5, bias=bias, act="silu", ft="bf16")
    assert_close(y.float().cpu().numpy(), bf16_round(pre + res), "bf16", what=f"tail split STD ({parts} parts)", pre=pre)
    # f32 hidden-stream update
    h = rng.normal(0, 1.5, (M, N)).astype(np.float32)
    h2 = ops.fused_gemm_addto(xd, pw, to_dev(h), sc, M=M)
    ref = h + gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, round_out=False)
    np.testing.assert_allclose(h2.cpu().numpy(), ref, rtol=2e-5, atol=3e-5 * np.abs(ref).max())
    # SwiGLU pair (128 output columns per workgroup: its own grid, its own tail)
    Ns = {4608: 4224, 4600: 4220, 7400: 4224}[N]   # 33 column blocks of 128 x 16 (or 9) row blocks
    parts2 = int(lib().dihip_gemm_prefill_tail_parts(wbits, M, Ns, K, G, 1))
    _, qg, sg, zg = make_case(rng, 1, Ns, K, G, wbits, "bf16", style="iq")
    _, qu, su, zu = make_case(rng, 1, Ns, K, G, wbits, "bf16", style="iq")
    pg = ops.pack_lowp(to_dev(qg), to_dev(sg, "bf16"), to_dev(zg, "bf16"), G, wbits)
    pu = ops.pack_lowp(to_dev(qu), to_dev(su, "bf16"), to_dev(zu, "bf16"), G, wbits)
    sc2 = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, Ns, K, G))
    act = ops.prenorm_swiglu(xd, pg, pu, sc2, M)
    g = gemm_ref.gemm_a16wx(x, qg, sg, zg, G, wbits, round_out=False).astype(np.float64)
    u = gemm_ref.gemm_a16wx(x, qu, su, zu, G, wbits, round_out=False).astype(np.float64)
    assert_close(act.float().cpu().numpy(), bf16_round(((g / (1 + np.exp(-g))) * u).astype(np.float32)), "bf16", what=f"tail split SwiGLU ({parts2} parts)")
    print(f"\n[prefill tail split] W{wbits} g{G} M{M} N{N} K{K}: {parts} parts per tail tile; SwiGLU pair N{Ns}: {parts2}")
