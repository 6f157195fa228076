"""Pin the GEMM oracle (numpy + plain C) against the REFERENCE's own host loops
(CPU_SubC_Ref / CPU_PerC_Ref / CPU_FP16W4_PerC_Ref / PackU8ToU4x2 /
CPU_Quant_Weight_*), compiled from /root/reference into oracle/_ref by oracle/Makefile.
Shapes follow tests/cpp/operator/cuda/operator_gemm_lowp_test.cpp:1034-1115 scaled down
so the CPU suite stays fast; data distributions follow :486-489.  CPU only."""
import numpy as np
import pytest

from oracle import cbind, gemm_ref, quant
from oracle.numerics import bf16_round, check_equal

needs_ref = pytest.mark.skipif(cbind.reflib() is None, reason="oracle/_ref/libdashinfer_ref.so not built")


def _ref_style_inputs(rng, M, N, K, G, wbits, ft):
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    srange = 2.0 / 256
    S = rng.uniform(srange * 0.9, srange * 1.1, ((K + G - 1) // G if G > 0 else 1, N)).astype(np.float32)
    Z = rng.uniform(-10, 10, S.shape).astype(np.float32)
    if wbits == 8:
        B = rng.integers(-1, 2, (K, N)).astype(np.int8)
    else:
        B = rng.integers(0, 3, (K, N)).astype(np.uint8)
    if ft == "bf16":
        A, S, Z = bf16_round(A), bf16_round(S), bf16_round(Z)
    return A, B, S, Z


@needs_ref
@pytest.mark.parametrize("ft", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K,G", [(1, 64, 256, -1), (3, 40, 200, 64), (17, 72, 130, 128), (2, 33, 96, 32)])
def test_c_oracle_bit_exact_vs_reference_loops_w8(M, N, K, G, ft):
    rng = np.random.default_rng(M * 1000 + N)
    A, B, S, Z = _ref_style_inputs(rng, M, N, K, G, 8, ft)
    alpha = 0.75
    ref = cbind.ref_gemm_a16w8(A, B, S, Z, G, alpha, ft)
    mine = cbind.gemm_a16wx(A, B, S, Z, G, 8, alpha=alpha, ft=ft)
    np.testing.assert_array_equal(mine, ref)  # same sequential f32 loop -> bit exact
    exact = gemm_ref.gemm_a16wx(A, B, S, Z, G, 8, alpha=alpha, ft=ft)
    assert check_equal(ref, exact) <= (1e-5 if ft == "f32" else 2 ** -7)


@needs_ref
@pytest.mark.parametrize("ft", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K,G", [(1, 64, 256, -1), (5, 34, 192, 64), (31, 48, 256, 128)])
def test_c_oracle_bit_exact_vs_reference_loops_w4(M, N, K, G, ft):
    rng = np.random.default_rng(7 + M)
    A, Bu, S, Z = _ref_style_inputs(rng, M, N, K, G, 4, ft)
    ref = cbind.ref_gemm_a16w4_unpacked(A, Bu, S, Z, G, 1.0, ft)
    packed = quant.pack_u4(Bu)
    np.testing.assert_array_equal(packed, cbind.ref_pack_u8_to_u4x2(Bu))
    mine = cbind.gemm_a16wx(A, packed, S, Z, G, 4, alpha=1.0, ft=ft)
    np.testing.assert_array_equal(mine, ref)


@needs_ref
def test_packed_perc_reference_loop():
    rng = np.random.default_rng(3)
    M, N, K = 4, 37, 160  # odd N exercises the nibble guard
    A, Bu, S, Z = _ref_style_inputs(rng, M, N, K, -1, 4, "f32")
    packed = quant.pack_u4(Bu)
    ref = cbind.ref_gemm_a16w4_perc_packed(A, packed, S, Z, N)
    mine = cbind.gemm_a16wx(A, packed, S, Z, -1, 4, ft="f32")
    np.testing.assert_array_equal(mine, ref)


@needs_ref
@pytest.mark.parametrize("G,qbits", [(-1, 8), (64, 8), (128, 4), (-1, 4)])
def test_test_side_quantiser_matches_reference(G, qbits):
    rng = np.random.default_rng(11)
    K, N = 256, 24
    W = rng.uniform(-1, 1, (K, N)).astype(np.float32)
    qmin, qmax = ((-128.0, 127.0) if qbits == 8 else (0.0, 15.0))
    q, s, z = quant.test_quant_weight(W, G, qmin, qmax, ft="f32")
    rq, rs, rz = cbind.ref_test_quant_weight(W, G, qbits)
    np.testing.assert_array_equal(q, rq)
    np.testing.assert_array_equal(s, rs)
    np.testing.assert_array_equal(z, rz)


def test_numpy_vs_c_oracle_modes():
    rng = np.random.default_rng(5)
    M, N, K, G = 3, 48, 256, 128
    W = bf16_round(rng.normal(0, 0.02, (K, N)))
    x = bf16_round(rng.uniform(-1, 1, (M, K)))
    bias = bf16_round(rng.normal(0, 0.02, N))
    for wbits, qf in ((8, quant.iq_quantize_a16w8), (4, quant.iq_quantize_a16w4)):
        q, s, z = qf(W, G, "bf16")
        for act in (None, "silu", "relu"):
            a = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, alpha=1.0, bias=bias, act=act, ft="bf16")
            c = cbind.gemm_a16wx(x, q, s, z, G, wbits, alpha=1.0, bias=bias, act=act, ft="bf16")
            assert check_equal(a, c) <= 2 ** -7
        xb = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, mode="x86_bf16")
        cb = cbind.gemm_a16wx(x, q, s, z, G, wbits, x86bf16=True)
        np.testing.assert_allclose(xb, cb, rtol=2e-5, atol=2e-6)
        # the two precisions of the x86 path agree to bf16 accuracy
        full = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, round_out=False)
        assert np.abs(full - xb).max() <= 1e-2 * max(1.0, np.abs(full).max())
