"""f16 activations on the FRAG32 small-batch kernels (K-slice, panel, gemv_batch) and on the context-phase GEMM (round 5; until
round 4 these were bf16-only and an f16 model at 4 < M <= 32 or in prefill fell back to the general kernel).  The reference serves
fp16 and bf16 alike (csrc/core/kernel/cuda/gemm_lowp/gemm_a16w4_kernel.h:71-131).  Against the numpy oracle with FT = f16
(oracle/gemm_ref.py), the tolerance of test_gpu_gemm.py: one f16 ulp (2^-10 relative) + noise floor."""
import os

import numpy as np
import pytest
import torch

from oracle import gemm_ref, glue
from oracle.numerics import f16_round
from tests.test_gpu_gemm import assert_close, make_case, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(pkg):
    from dash_infer_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("wbits,G", [(4, 128), (8, -1), (8, 128), (4, 256)])
@pytest.mark.parametrize("M", [7, 17, 32])
def test_small_batch_forms_f16(ops, wbits, G, M):
    rng = np.random.default_rng(M * 13 + wbits + G)
    K, N = 4096, 272
    x, q, s, z = make_case(rng, M, N, K, G, wbits, "f16")
    _, q2, s2, z2 = make_case(rng, 1, N, K, G, wbits, "f16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "f16"), to_dev(z, "f16"), G, wbits)
    pw2 = ops.pack_lowp(to_dev(q2), to_dev(s2, "f16"), to_dev(z2, "f16"), G, wbits)
    bias = f16_round(rng.normal(0, 0.5, N).astype(np.float32))
    xd = to_dev(x, "f16")
    y = ops.gemm_lowp(xd, pw, bias=to_dev(bias, "f16"), act="silu", alpha=0.5)
    ref = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, alpha=0.5, bias=bias, act="silu", ft="f16")
    assert_close(y.float().cpu().numpy(), ref, "f16", what="std")
    sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
    h = rng.normal(0, 1, (M, N)).astype(np.float32)
    out = ops.fused_gemm_addto(xd, pw, torch.from_numpy(h).cuda(), sc)
    ref2 = h + gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, ft="f32")
    np.testing.assert_allclose(out.cpu().numpy(), ref2, rtol=1e-3, atol=1e-3 * np.abs(ref2).max())
    hk = rng.normal(0, 1.5, (M, K)).astype(np.float32)
    gamma = f16_round(rng.normal(1, 0.1, K).astype(np.float32))
    frag = bool(ops.prefers_frag(pw, M, dual=True)) and M > 4
    act = ops.fused_norm_swiglu(torch.from_numpy(hk).cuda(), to_dev(gamma, "f16"), 1e-6, pw, pw2, sc,
                                y_layout=ops.ACT_FRAG32 if frag else ops.ACT_ROWMAJOR)
    if frag:
        act = ops.act_from_frag(act, M, N)
    xn = f16_round(glue.rmsnorm(hk, gamma, 1e-6))
    ref3 = f16_round(glue.silu(gemm_ref.gemm_a16wx(xn, q, s, z, G, wbits, ft="f32")) * gemm_ref.gemm_a16wx(xn, q2, s2, z2, G, wbits, ft="f32"))
    assert_close(act.float().cpu().numpy(), ref3, "f16", what="swiglu", pre=ref3)


@pytest.mark.parametrize("wbits,G,M", [(4, 128, 16), (4, 128, 32), (8, -1, 32)])
def test_full_size_mlp_f16_kslice_and_panel(ops, wbits, G, M):
    """the 7B MLP at full size in f16 through the FRAG32 chain: K-slice kernel (forced) against the panel kernel and the f64 oracle"""
    rng = np.random.default_rng(5 * M + wbits)
    K, I = 3584, 18944
    hk = torch.from_numpy(rng.normal(0, 1.0, (M, K)).astype(np.float32)).cuda()
    gamma = to_dev(f16_round(rng.normal(1, 0.1, K).astype(np.float32)), "f16")
    _, qg, sg, zg = make_case(rng, 1, I, K, G, wbits, "f16")
    _, qu, su, zu = make_case(rng, 1, I, K, G, wbits, "f16")
    _, qd, sd, zd = make_case(rng, 1, K, I, G, wbits, "f16")
    pg, pu, pd = (ops.pack_lowp(to_dev(q_), to_dev(s_, "f16"), to_dev(z_, "f16"), G, wbits) for q_, s_, z_ in ((qg, sg, zg), (qu, su, zu), (qd, sd, zd)))
    sc = ops.Scratch(max(ops.lowp_workspace_bytes(wbits, M, I, K, G), ops.lowp_workspace_bytes(wbits, M, K, I, G)))
    h0 = torch.from_numpy(rng.normal(0, 1, (M, K)).astype(np.float32)).cuda()
    old = os.environ.get("DIHIP_GEMM_KSLICE")

    def run(mode):
        os.environ["DIHIP_GEMM_KSLICE"] = mode
        act = ops.fused_norm_swiglu(hk, gamma, 1e-6, pg, pu, sc, y_layout=ops.ACT_FRAG32)
        out = ops.fused_gemm_addto(act, pd, h0, sc, x_layout=ops.ACT_FRAG32, M=M)
        return ops.act_from_frag(act, M, I), out

    try:
        a_pan, o_pan = run("0")
        a_ksl, o_ksl = run("2")
    finally:
        if old is None:
            os.environ.pop("DIHIP_GEMM_KSLICE", None)
        else:
            os.environ["DIHIP_GEMM_KSLICE"] = old
    ap, ak = a_pan.float().cpu().numpy(), a_ksl.float().cpu().numpy()
    assert_close(ak, ap, "f16", what="swiglu: K-slice vs panel kernel", pre=ap)
    cols = rng.choice(I, 48, replace=False)
    xn = f16_round(glue.rmsnorm(hk.cpu().numpy(), gamma.float().cpu().numpy(), 1e-6)).astype(np.float64)
    ref_a = glue.silu(xn @ gemm_ref.dequant(qg, sg, zg, G, wbits)[:, cols].astype(np.float64)) * \
        (xn @ gemm_ref.dequant(qu, su, zu, G, wbits)[:, cols].astype(np.float64))
    np.testing.assert_allclose(ak[:, cols], ref_a, rtol=2e-3, atol=2e-3 * np.abs(ref_a).max())
    cols = rng.choice(K, 48, replace=False)
    h0n = h0.cpu().numpy()
    for o_, a_ in ((o_ksl, ak), (o_pan, ap)):
        ref_o = h0n[:, cols] + a_.astype(np.float64) @ gemm_ref.dequant(qd, sd, zd, G, wbits)[:, cols].astype(np.float64)
        np.testing.assert_allclose(o_.cpu().numpy()[:, cols], ref_o, rtol=2e-3, atol=2e-3 * np.abs(ref_o).max())


@pytest.mark.parametrize("wbits,G", [(4, 128), (8, -1), (8, 64)])
@pytest.mark.parametrize("M,N,K", [(64, 512, 512), (200, 261, 1024), (2048, 640, 3584)])
def test_prefill_gemm_f16(ops, M, N, K, wbits, G):
    rng = np.random.default_rng(M + N + wbits)
    x, q, s, z = make_case(rng, M, N, K, G, wbits, "f16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "f16"), to_dev(z, "f16"), G, wbits)
    bias = f16_round(rng.normal(0, 0.5, N).astype(np.float32))
    y = ops.gemm_lowp(to_dev(x, "f16"), pw, bias=to_dev(bias, "f16"))
    ref = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, bias=bias, ft="f16")
    assert_close(y.float().cpu().numpy(), ref, "f16", what="prefill std")
    sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
    h = rng.normal(0, 1, (M, N)).astype(np.float32)
    out = ops.fused_gemm_addto(to_dev(x, "f16"), pw, torch.from_numpy(h).cuda(), sc, M=M)
    ref2 = h + gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, ft="f32")
    np.testing.assert_allclose(out.cpu().numpy(), ref2, rtol=1e-3, atol=1e-3 * np.abs(ref2).max())
