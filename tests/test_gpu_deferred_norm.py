"""Deferred RMSNorm between a residual GEMM and the next GEMM of a batched decode layer (include/dashinfer_hip.h, "with the RMSNorm
DEFERRED"; reference operators: Gemm -> Binary ADD -> LayerNormNoBeta -> Gemm, qwen_v15.py:296-361, layernorm.cpp:110-157).

  producer  dihip_fused_gemm_addto_prenorm:  h_out BIT-IDENTICAL to dihip_fused_gemm_addto; xnorm == FT(gamma * h_out) exactly (one
            rounding of the f32 product); rowsq = per-workgroup partial sums of h_out^2 whose total is Sum_n h_out^2 to f32 rounding,
            zero for the rows past M;
  consumer  dihip_prenorm_swiglu_rowsq / dihip_prenorm_gemm_rowsq: the rounding point moves (FT(gamma * x) . W scaled by 1 / rms
            instead of FT(gamma * x / rms) . W), the relative precision must not.  Both forms -- the deferred one and the chain that
            normalises first (dihip_prenorm_* on the finished norm, which equals the oracle of the reference's operators to the
            last bit but libm) -- are scored against the UNROUNDED mathematics in float64 (norm, weight-only GEMM, SwiGLU / bias
            with no intermediate rounding): the deferred form's rms and maximum error must be those of the reference's own
            rounding (ratios printed -- measured 0.98 ... 1.01 and 0.8 ... 1.4; bounds 1.1 and 2).  About two thirds of the
            output elements differ from the normalise-first chain in the last bits: an element-wise comparison with that chain's
            oracle says nothing (both are ~1 bf16 ulp of the LARGEST output away from the mathematics, and small outputs have
            small ulps).
Shapes: the o-projection -> gate/up pair of Qwen2-7B (batch 32 / 17: K-slice kernel, MT = 2, helper wave) and of one TP = 8 rank of
Qwen2-72B (batch 16: K-slice kernel with a slab, the reduction kernel applies 1 / rms), and a narrow consumer served by the
whole-column kernel (plain epilogue + bias)."""
import numpy as np
import pytest
import torch

from oracle import gemm_ref
from oracle import glue
from tests.test_gpu_gemm import assert_close, bf16_round, make_case, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(pkg):
    from dash_infer_amd import ops as o
    return o


def _producer(ops, rng, M, N, K, G, wbits, consumer, dual):
    """o-projection-like residual GEMM [M, K] x [K, N] with the norm deferred for `consumer` -> everything the checks need"""
    x, q, s, z = make_case(rng, M, N, K, G, wbits, "bf16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
    sc = ops.Scratch(max(ops.lowp_workspace_bytes(wbits, M, N, K, G), ops.lowp_workspace_bytes(wbits, M, consumer.N, N, G)) * 2)
    h = torch.from_numpy(rng.normal(0, 1, (M, N)).astype(np.float32)).cuda()
    gamma = to_dev(bf16_round(rng.normal(1, 0.1, N).astype(np.float32)), "bf16")
    xd = to_dev(x, "bf16")
    frag = ops.prefers_frag(pw, M)
    xin = ops.act_to_frag(xd) if frag else xd
    lay = ops.ACT_FRAG32 if frag else ops.ACT_ROWMAJOR
    cfrag = ops.prefers_frag(consumer, M, dual=dual)
    clay = ops.ACT_FRAG32 if cfrag else ops.ACT_ROWMAJOR
    assert ops.prenorm_rowsq_supported(consumer, M, dual=dual, x_layout=clay), "the consumer's kernel must take deferred rows for this shape"
    ref_h = ops.fused_gemm_addto(xin, pw, h, sc, x_layout=lay, M=M)
    xg = torch.zeros(ops.act_frag_numel(M, N) if cfrag else M * N, dtype=torch.bfloat16, device="cuda")
    rowsq = ops.rowsq_buffer("cuda")
    rowsq.fill_(float("nan"))   # (every part the consumer reads must have been written)
    out_h, parts = ops.fused_gemm_addto_prenorm(xin, pw, h, sc, gamma, 1e-6, xg, rowsq, x_layout=lay, xnorm_layout=clay, M=M)
    return dict(pw=pw, sc=sc, h=h, gamma=gamma, xin=xin, lay=lay, clay=clay, cfrag=cfrag, ref_h=ref_h, out_h=out_h, parts=parts, xg=xg,
                rowsq=rowsq)


def _true_norm(h, gamma, eps):
    h64 = h.astype(np.float64)
    return (gamma.astype(np.float64)[None, :] * h64 / np.sqrt((h64 * h64).mean(1, keepdims=True) + eps)).astype(np.float32)


def _score(what, y_def, y_chain, ref_oracle, truth):
    """deferred form and normalise-first chain against the unrounded mathematics; the deferred form against the chain's oracle"""
    yd, yc = y_def.float().cpu().numpy().astype(np.float64), y_chain.float().cpu().numpy().astype(np.float64)
    ed, ec = yd - truth, yc - truth
    rms_d, rms_c = np.sqrt((ed * ed).mean()), np.sqrt((ec * ec).mean())
    max_d, max_c = np.abs(ed).max(), np.abs(ec).max()
    scale = np.abs(truth).max()
    eo = ref_oracle.astype(np.float64) - truth   # (the oracle of the reference's operators: the chain's error, libm aside)
    print(f"\n[deferred norm: {what}] error against float64 mathematics at output scale {scale:.3g}: rms {rms_d:.3e} (chain that normalises "
          f"first: {rms_c:.3e}, ratio {rms_d / rms_c:.3f}), max {max_d:.3e} (chain {max_c:.3e}, ratio {max_d / max_c:.2f}); elements differing "
          f"from the chain's output: {float((yd != yc).mean()):.1%}; the oracle of the reference's chain: rms {np.sqrt((eo * eo).mean()):.3e}")
    assert rms_d <= 1.1 * rms_c, "the deferred form is less precise than the reference's rounding"
    assert max_d <= 2.0 * max_c + 1e-6
    assert np.sqrt((eo * eo).mean()) <= 1.02 * rms_c and rms_c <= 1.02 * np.sqrt((eo * eo).mean()), "the normalise-first chain left its oracle"


def _check_producer(ops, st, M, N):
    assert st["parts"] > 0, "the whole-column kernel serves this residual GEMM: it must offer the deferred form"
    assert torch.equal(st["out_h"], st["ref_h"]), "h_out differs from dihip_fused_gemm_addto"
    xg_rm = ops.act_from_frag(st["xg"], M, N) if st["cfrag"] else st["xg"].view(M, N)
    want = (st["gamma"].float()[None, :] * st["ref_h"]).bfloat16()
    assert torch.equal(xg_rm.view(torch.int16), want.view(torch.int16)), "xnorm != bf16(gamma * h_out)"
    rs = st["rowsq"][: st["parts"] * 32].view(st["parts"], 32)
    assert torch.isfinite(rs).all()
    tot = rs.double().sum(0).cpu().numpy()
    ref = (st["ref_h"].double() ** 2).sum(1).cpu().numpy()
    np.testing.assert_allclose(tot[:M], ref, rtol=2e-6)
    assert (tot[M:] == 0).all(), "rows past M must read as zero"


@pytest.mark.parametrize("M", [32, 17, 16, 5])
@pytest.mark.parametrize("shape", ["qwen2_7b", "qwen2_72b_tp8_rank"])
def test_o_projection_to_swiglu_pair(ops, shape, M):
    hidden, attn_w, inter = (3584, 3584, 18944) if shape == "qwen2_7b" else (8192, 1024, 3712)
    wbits, G = 4, 128
    rng = np.random.default_rng(M * 7 + len(shape))
    _, q1, s1, z1 = make_case(rng, 1, inter, hidden, G, wbits, "bf16")
    _, q2, s2, z2 = make_case(rng, 1, inter, hidden, G, wbits, "bf16")
    p1 = ops.pack_lowp(to_dev(q1), to_dev(s1, "bf16"), to_dev(z1, "bf16"), G, wbits)
    p2 = ops.pack_lowp(to_dev(q2), to_dev(s2, "bf16"), to_dev(z2, "bf16"), G, wbits)
    st = _producer(ops, rng, M, hidden, attn_w, G, wbits, p1, True)
    _check_producer(ops, st, M, hidden)
    sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, inter, hidden, G) * 2)
    y_def = ops.prenorm_swiglu_rowsq(st["xg"], p1, p2, sc, M, st["rowsq"], st["parts"], 1e-6, x_layout=st["clay"])
    # the chain that normalises first
    xn = torch.zeros_like(st["xg"])
    ops.fused_gemm_addto_norm(st["xin"], st["pw"], st["h"], st["sc"], st["gamma"], 1e-6, xn, x_layout=st["lay"], xnorm_layout=st["clay"], M=M)
    y_chain = ops.prenorm_swiglu(xn, p1, p2, sc, M, x_layout=st["clay"])
    # the oracle of the reference's operators: LayerNormNoBeta -> FT -> weight-only GEMMs -> SwiGLU
    hn, gn = st["ref_h"].cpu().numpy(), st["gamma"].float().cpu().numpy()
    xn_ref = bf16_round(glue.rmsnorm(hn, gn, 1e-6))
    g_ = gemm_ref.gemm_a16wx(xn_ref, q1, s1, z1, G, wbits, ft="f32")
    u_ = gemm_ref.gemm_a16wx(xn_ref, q2, s2, z2, G, wbits, ft="f32")
    ref = bf16_round(glue.silu(g_) * u_)
    # the unrounded mathematics
    xt = _true_norm(hn, gn, 1e-6)
    gt = gemm_ref.gemm_a16wx(xt, q1, s1, z1, G, wbits, round_out=False).astype(np.float64)
    ut = gemm_ref.gemm_a16wx(xt, q2, s2, z2, G, wbits, round_out=False).astype(np.float64)
    truth = (gt / (1.0 + np.exp(-gt))) * ut
    _score(f"{shape}, M = {M}, SwiGLU pair, parts {st['parts']}", y_def, y_chain, ref, truth)


@pytest.mark.parametrize("M", [32, 9])
def test_down_like_producer_to_qkv_consumer_on_the_whole_column_kernel(ops, M):
    """a residual GEMM served by the whole-column kernel feeding a plain GEMM + bias (the qkv projection's form)"""
    hidden, K, Nq = 3584, 1024, 4608
    wbits, G = 4, 128
    rng = np.random.default_rng(M + 99)
    _, q1, s1, z1 = make_case(rng, 1, Nq, hidden, G, wbits, "bf16")
    p1 = ops.pack_lowp(to_dev(q1), to_dev(s1, "bf16"), to_dev(z1, "bf16"), G, wbits)
    bias = bf16_round(rng.normal(0, 0.5, Nq).astype(np.float32))
    st = _producer(ops, rng, M, hidden, K, G, wbits, p1, False)
    _check_producer(ops, st, M, hidden)
    sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, Nq, hidden, G) * 2)
    y_def = ops.prenorm_gemm_rowsq(st["xg"], p1, to_dev(bias, "bf16"), sc, M, st["rowsq"], st["parts"], 1e-6, x_layout=st["clay"])
    xn = torch.zeros_like(st["xg"])
    ops.fused_gemm_addto_norm(st["xin"], st["pw"], st["h"], st["sc"], st["gamma"], 1e-6, xn, x_layout=st["lay"], xnorm_layout=st["clay"], M=M)
    y_chain = ops.prenorm_gemm(xn, p1, to_dev(bias, "bf16"), sc, M, x_layout=st["clay"])
    hn, gn = st["ref_h"].cpu().numpy(), st["gamma"].float().cpu().numpy()
    xn_ref = bf16_round(glue.rmsnorm(hn, gn, 1e-6))
    ref = gemm_ref.gemm_a16wx(xn_ref, q1, s1, z1, G, wbits, bias=bias, ft="bf16")
    truth = gemm_ref.gemm_a16wx(_true_norm(hn, gn, 1e-6), q1, s1, z1, G, wbits, round_out=False).astype(np.float64) + bias.astype(np.float64)
    _score(f"M = {M}, GEMM + bias, parts {st['parts']}", y_def, y_chain, ref, truth)


def test_refusals(ops):
    """shapes / types no deferred form exists for are reported as such, and asking anyway fails loudly"""
    rng = np.random.default_rng(3)
    _, q, s, z = make_case(rng, 1, 1024, 512, 128, 4, "bf16")
    pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), 128, 4)
    assert not ops.prenorm_rowsq_supported(pw, 1)        # batch 1: the M <= 4 kernel normalises in its prologue
    assert not ops.prenorm_rowsq_supported(pw, 64)       # context phase
    assert not ops.prenorm_rowsq_supported(pw, 16, dual=True)   # SwiGLU pair on the whole-column kernel
    sc = ops.Scratch(ops.lowp_workspace_bytes(4, 64, 1024, 512, 128))
    x = torch.zeros(64, 512, dtype=torch.bfloat16, device="cuda")
    rowsq = ops.rowsq_buffer("cuda")
    with pytest.raises(Exception):
        ops.prenorm_gemm_rowsq(x, pw, None, sc, 64, rowsq, 4, 1e-6)
