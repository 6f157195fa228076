"""GPU parity of the MFMA prefill attention (dihip_prefill_attn, through the C-ABI).

Oracle: oracle/attention.py:prefill_attention (restates the host loop of
tests/cpp/kernel/cuda/kernel_mhaprefill_test.cpp:119-250, generalised to GQA and a cached prefix)
and, for the MHA / no-prefix layout the reference's own test uses, the reference's checker
pefill_check_with_reference compiled into oracle/_ref (feps 1e-3 for f16 as in the reference test;
bf16 gets the north-star tolerance 1e-2).  Cases: ragged lengths around the 64-row / 32-key tile
edges, GQA, a cached prefix (MIX format: q rows of the fused tensor, contiguous K/V), the
INTERLEAVED format (q, k, v all views of the fused qkv rows), the BASELINE shape (Qwen2-7B heads,
2048 tokens) through size-independent properties.
"""
import numpy as np
import pytest
import torch

from oracle import attention, cbind
from oracle.numerics import bf16_round, f16_round

pytestmark = pytest.mark.gpu
TD = {"bf16": torch.bfloat16, "f16": torch.float16}
RND = {"bf16": bf16_round, "f16": f16_round}
TOL = {"bf16": 1e-2, "f16": 2e-3}


@pytest.fixture(scope="module")
def ops(pkg):
    from dash_infer_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def dev(a, ft):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(TD[ft]).cuda()


@pytest.mark.parametrize("ft", ["bf16", "f16"])
@pytest.mark.parametrize("Lq,Lk,n,g", [(1, 1, 2, 2), (5, 5, 4, 2), (63, 63, 2, 1), (64, 64, 4, 4), (65, 97, 4, 2),
                                       (130, 130, 14, 2), (200, 456, 8, 8), (33, 1000, 7, 1)])
def test_prefill_matches_oracle(ops, Lq, Lk, n, g, ft):
    rng = np.random.default_rng(Lq * 7 + Lk)
    H = 128
    q = RND[ft](rng.normal(0, 1, (Lq, n, H)).astype(np.float32))
    k = RND[ft](rng.normal(0, 1, (Lk, g, H)).astype(np.float32))
    v = RND[ft](rng.normal(0, 1, (Lk, g, H)).astype(np.float32))
    alpha = 1.0 / np.sqrt(H)
    ref = attention.prefill_attention(q, k, v, alpha)
    out = ops.prefill_attn(dev(q.reshape(Lq, -1), ft), dev(k.reshape(Lk, -1), ft), dev(v.reshape(Lk, -1), ft), n, g, H, alpha)
    np.testing.assert_allclose(out.float().cpu().numpy().reshape(Lq, n, H), ref, rtol=TOL[ft], atol=TOL[ft] * 0.5)  # P is rounded to FT for the matrix core


def test_prefill_interleaved_qkv_rows_and_non_causal(ops):
    """INTERLEAVED format (span_attn_op_cuda.cpp:489-502): q, k, v are column slices of the fused
    [L, (n+2g)*H] rows; also the non-causal switch."""
    rng = np.random.default_rng(3)
    L, n, g, H, ft = 150, 6, 2, 128, "bf16"
    qkv = bf16_round(rng.normal(0, 1, (L, (n + 2 * g) * H)).astype(np.float32))
    t = dev(qkv, ft)
    q, k, v = t[:, : n * H], t[:, n * H:(n + g) * H], t[:, (n + g) * H:]
    alpha = 0.1
    for causal in (True, False):
        ref = attention.prefill_attention(qkv[:, : n * H].reshape(L, n, H), qkv[:, n * H:(n + g) * H].reshape(L, g, H),
                                          qkv[:, (n + g) * H:].reshape(L, g, H), alpha, causal)
        out = ops.prefill_attn(q, k, v, n, g, H, alpha, causal)
        np.testing.assert_allclose(out.float().cpu().numpy().reshape(L, n, H), ref, rtol=1e-2, atol=2.5e-3)


def test_prefill_passes_the_reference_checker(ops):
    """The reference's own host check (kernel_mhaprefill_test.cpp:119-320) on its own layout:
    concat [batch, seqlen, 3, nhead, phead], f16, feps 1e-3."""
    if cbind.reflib() is None:
        pytest.skip("oracle/_ref/libdashinfer_ref.so not built")
    rng = np.random.default_rng(11)
    B, L, nh, H = 2, 77, 3, 128
    concat = f16_round(rng.uniform(-1, 1, (B, L, 3, nh, H)).astype(np.float32))
    alpha = 1.0 / np.sqrt(H)
    out = np.empty((B, L, nh, H), np.float32)
    for b in range(B):
        t = dev(concat[b].reshape(L, 3 * nh * H), "f16")
        # rows are [q heads | k heads | v heads]
        o = ops.prefill_attn(t[:, : nh * H], t[:, nh * H: 2 * nh * H], t[:, 2 * nh * H:], nh, nh, H, alpha)
        out[b] = o.float().cpu().numpy().reshape(L, nh, H)
    assert cbind.ref_prefill_check(concat, out, alpha, True, 1e-3)


def test_prefill_baseline_shape_properties(ops):
    """Qwen2-7B heads (n=28, g=4), 2048 tokens: (i) the first rows equal a short-sequence call
    (causality), (ii) row 0 equals V[0] of its KV head, (iii) a 64-token suffix computed over the
    cached prefix (MIX format, Lq < Lk) equals the same rows of the full call, (iv) spot rows
    against the oracle."""
    rng = np.random.default_rng(5)
    L, n, g, H, ft = 2048, 28, 4, 128, "bf16"
    q = bf16_round(rng.normal(0, 1, (L, n, H)).astype(np.float32))
    k = bf16_round(rng.normal(0, 1, (L, g, H)).astype(np.float32))
    v = bf16_round(rng.normal(0, 1, (L, g, H)).astype(np.float32))
    alpha = 1.0 / np.sqrt(H)
    qd, kd, vd = dev(q.reshape(L, -1), ft), dev(k.reshape(L, -1), ft), dev(v.reshape(L, -1), ft)
    full = ops.prefill_attn(qd, kd, vd, n, g, H, alpha)
    short = ops.prefill_attn(qd[:100].contiguous(), kd[:100].contiguous(), vd[:100].contiguous(), n, g, H, alpha)
    assert torch.equal(full[:64], short[:64])          # same tiles, same arithmetic
    np.testing.assert_allclose(full[:100].float().cpu().numpy(), short.float().cpu().numpy(), rtol=1e-2, atol=2.5e-3)
    np.testing.assert_array_equal(full[0].float().cpu().numpy().reshape(n, H), np.repeat(v[0], n // g, axis=0))
    tail = ops.prefill_attn(qd[-64:].contiguous(), kd, vd, n, g, H, alpha)
    np.testing.assert_allclose(tail.float().cpu().numpy(), full[-64:].float().cpu().numpy(), rtol=1e-2, atol=2.5e-3)
    rows = [1, 63, 64, 1000, 2047]
    ref = attention.prefill_attention(q, k, v, alpha)[rows]
    np.testing.assert_allclose(full[rows].float().cpu().numpy().reshape(len(rows), n, H), ref, rtol=1e-2, atol=2.5e-3)


def test_prefill_error_behaviour(ops):
    from dash_infer_amd import capi
    t = torch.zeros(4, 256, dtype=torch.bfloat16, device="cuda")
    rc = ops.lib().dihip_prefill_attn(ops.cur_stream(), ops.ptr(t), ops.ptr(t), ops.ptr(t), ops.ptr(t), 4, 4, 256, 256, 4, 2, 64, 1,
                                      1.0, capi.BF16)
    assert rc == capi.PARAM_ERROR and b"head size" in ops.lib().dihip_last_error()
    rc = ops.lib().dihip_prefill_attn(ops.cur_stream(), ops.ptr(t), ops.ptr(t), ops.ptr(t), ops.ptr(t), 0, 0, 256, 256, 2, 2, 128, 1,
                                      1.0, capi.BF16)
    assert rc == capi.SUCCESS  # empty prefill
    rc = ops.lib().dihip_prefill_attn(ops.cur_stream(), ops.ptr(t), ops.ptr(t), ops.ptr(t), ops.ptr(t), 2, 2, 256, 256, 2, 2, 128, 1,
                                      0.0, capi.BF16)
    assert rc == capi.PARAM_ERROR and b"scale" in ops.lib().dihip_last_error()       # exp2 folding needs alpha > 0
    rc = ops.lib().dihip_prefill_attn(ops.cur_stream(), ops.ptr(t), ops.ptr(t), ops.ptr(t), ops.ptr(t), 2, 1 << 20, 256, 4096, 2, 2,
                                      128, 1, 1.0, capi.BF16)
    assert rc == capi.EXCEED_LIMIT_ERROR and b"2 GiB" in ops.lib().dihip_last_error()   # 32-bit buffer offsets


@pytest.mark.gpu
@pytest.mark.parametrize("causal,ft", [(True, "bf16"), (False, "bf16"), (True, "f16")])
def test_prefill_two_key_groups(ops, causal, ft):
    """Grids between one and two workgroups per CU run the 8-wave form: two 4-wave groups take alternate key tiles of a
    query tile and merge (O, m, l) through LDS.  5 x 64 = 320 workgroups, ragged lengths, seq_k > seq_q."""
    rng = np.random.default_rng(11)
    Lq, Lk, n, g, H = 600, 700, 64, 8, 128
    q = RND[ft](rng.normal(0, 1, (Lq, n, H)).astype(np.float32))
    k = RND[ft](rng.normal(0, 1, (Lk, g, H)).astype(np.float32))
    v = RND[ft](rng.normal(0, 1, (Lk, g, H)).astype(np.float32))
    alpha = 1.0 / np.sqrt(H)
    ref = attention.prefill_attention(q, k, v, alpha, causal)
    out = ops.prefill_attn(dev(q.reshape(Lq, -1), ft), dev(k.reshape(Lk, -1), ft), dev(v.reshape(Lk, -1), ft), n, g, H, alpha, causal)
    np.testing.assert_allclose(out.float().cpu().numpy().reshape(Lq, n, H), ref, rtol=TOL[ft], atol=TOL[ft] * 0.5)


@pytest.mark.gpu
def test_prefill_full_size_2048_properties(ops):
    """BASELINE sequence length at the Qwen2-7B head counts (448 workgroups: the two-key-group form): the first row
    attends to key 0 only, and the first 128 rows agree with a 128-token call (another tile shape)."""
    rng = np.random.default_rng(12)
    L, n, g, H = 2048, 28, 4, 128
    qkv = torch.from_numpy(rng.normal(0, 1, (L, (n + 2 * g) * H)).astype(np.float32)).to(torch.bfloat16).cuda()
    q, k, v = qkv[:, : n * H], qkv[:, n * H:(n + g) * H], qkv[:, (n + g) * H:]
    full = ops.prefill_attn(q, k, v, n, g, H, 0.088)
    assert torch.isfinite(full.float()).all()
    v0 = v[0].float().cpu().numpy().reshape(g, H)
    np.testing.assert_array_equal(full[0].float().cpu().numpy().reshape(n, H), np.repeat(v0, n // g, axis=0))
    short = ops.prefill_attn(q[:128], k[:128], v[:128], n, g, H, 0.088)
    np.testing.assert_allclose(full[:128].float().cpu().numpy(), short.float().cpu().numpy(), rtol=1e-2, atol=2.5e-3)
    # last row against the f64 oracle (one query over all 2048 keys)
    qn = q[-1:].float().cpu().numpy().reshape(1, n, H)
    ref = attention.prefill_attention(qn, k.float().cpu().numpy().reshape(L, g, H), v.float().cpu().numpy().reshape(L, g, H), 0.088)
    np.testing.assert_allclose(full[-1].float().cpu().numpy().reshape(1, n, H), ref, rtol=1e-2, atol=2.5e-3)
