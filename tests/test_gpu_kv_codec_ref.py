"""The quantised-KV codec pinned against the REFERENCE'S OWN DEVICE CODE (the reference's tests pin only
QuantMode::NONE, tests/cpp/.../test_quant_none.cpp).

oracle/_ref/libdashinfer_ref_codec_{ieee,rcp}.so are span-attention/src/cache_quant/impl_{i8,u4}.cuh (+ reduce / utils / prototype /
config) compiled for gfx950 from where they lie (oracle/Makefile `refcodec`, stand-ins for the PTX / CUDA headers in
oracle/ref_cuda_shim), driven the way the reference's append kernel drives them (oracle/codec_ref.hip: one 32-lane warp per head
row, Builder -> Quant, decoder_cache_append.cuh:33-86).  The product's writers -- dihip_kv_context_copy (prefill rows) and
dihip_kv_append (decode step) -- must store the SAME bytes and the same {zero, scale} pair for every (token, head), random rows and
the adversarial heads alike.

__fdividef (cache_quant/utils.cuh:28-32): hipcc's own __fdividef is IEEE division; CUDA's is a <= 2-ulp approximation whose bits
cannot be reproduced off NVIDIA hardware.  The second library divides as a * v_rcp_f32(b) -- an approximate division of that error
class -- and the test MEASURES what such a division can move: scales within 2 ulp, zero-points by at most 1 (a rint boundary),
codes by at most 2, on a bounded fraction of the elements (printed; profiles/r05_kv_codec_ref_pin.txt).  The product divides exactly (as oracle/kv_codec.py)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODE = {"i8": 1, "u4": 2}
HB = {"i8": 128, "u4": 64}


def _ref(division):
    path = os.path.join(ROOT, "oracle", "_ref", f"libdashinfer_ref_codec_{division}.so")
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (oracle/Makefile refcodec needs /root/reference; the prebuilt library travels with gpurun)")
    lib = C.CDLL(path)
    lib.ref_codec_quantize.restype = C.c_int
    lib.ref_codec_quantize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.ref_codec_division.restype = C.c_char_p
    return lib


def _ref_quantize(lib, mode, rows_f32):
    """rows_f32: torch f32 [R, 128] on the GPU -> (bytes [R, HB], params f32 [R, 2] = {zero, scale})"""
    R = rows_f32.shape[0]
    q = torch.empty(R, HB[mode], dtype=torch.uint8, device="cuda")
    prm = torch.empty(R, 2, dtype=torch.float32, device="cuda")
    rc = lib.ref_codec_quantize(MODE[mode], rows_f32.data_ptr(), q.data_ptr(), prm.data_ptr(), R, None)
    assert rc == 0
    torch.cuda.synchronize()
    return q, prm


def _heads(ft, L, g, seed):
    """[L, g, 128] f32 values exactly representable in ft: random rows of several scales + the adversarial heads"""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(L, g, 128, generator=gen) * torch.tensor([0.02, 0.5, 1.0, 3.0, 40.0, 1e-4])[torch.randint(0, 6, (L, g, 1), generator=gen)]
    x[1, 0] = -x[1, 0].abs() - 1.0                       # all-negative: the u4 zero clamps at 15, elements saturate at 0 (impl_u4.cuh:79-93)
    x[2, 0] = 0.5                                          # constant: scale clamps to EPS
    x[3, 0] = -0.75                                        # negative constant
    x[4, 0] = torch.linspace(-2, -1, 128)                  # the VERDICT r1 reproducer
    x[5, 0] = x[5, 0].abs() + 1.0                          # all-positive
    x[6, 0] = 0.0
    x[7, 0] = torch.arange(128) % 16 * 1.0                 # every u4 code exactly, zero = 0, scale = 1: ties impossible
    x[8, 0] = (torch.arange(128) % 16 * 1.0 + 0.5)         # ... and every value on a .5 tie of the u4 grid (rint: to even)
    x[9, 0] = (torch.arange(128) - 64) * 0.5               # i8 grid ties
    x[10, 0, 0] = 3.0e4                                    # one huge element (inside the f16 range)
    x[11, 0] = torch.cat([torch.full((64,), -1e-3), torch.full((64,), 1e-3)])
    dt = torch.bfloat16 if ft == "bf16" else torch.float16
    return x.to(dt).float()


@pytest.mark.parametrize("mode", ["u4", "i8"])
@pytest.mark.parametrize("ft", ["bf16", "f16"])
def test_product_span_bytes_equal_the_reference_device_codec(pkg, mode, ft):
    from dash_infer_amd import ops
    lib = _ref("ieee")
    assert lib.ref_codec_division() == b"ieee"
    dt = torch.bfloat16 if ft == "bf16" else torch.float16
    g, H, S, L, n = 4, 128, 32, 96, 4
    heads = _heads(ft, L, g, seed=len(mode) * 7 + len(ft))           # K rows; V = a permutation of them
    vheads = heads.flip(0)
    want = {}
    for name, hx in (("k", heads), ("v", vheads)):
        q, prm = _ref_quantize(lib, mode, hx.reshape(L * g, H).cuda().contiguous())
        want[name] = (q.cpu().numpy().reshape(L, g, HB[mode]), prm.cpu().numpy().reshape(L, g, 2))

    def check(pool, kv, upto, tag):
        hb = HB[mode]
        for name, idxs in (("k", kv.k_idx[0]), ("v", kv.v_idx[0])):
            wq, wp = want[name]
            for i, si in enumerate(idxs):
                valid = min(S, upto - i * S)
                if valid <= 0:
                    break
                raw = pool.span_view(si).cpu().numpy()
                data = raw[: g * S * hb].reshape(g, S, hb)[:, :valid]
                prm = raw[g * S * hb: g * S * hb + g * S * 8].view(np.float32).reshape(g, S, 2)[:, :valid]
                t0 = i * S
                np.testing.assert_array_equal(data, wq[t0:t0 + valid].transpose(1, 0, 2), err_msg=f"{tag} {name} span {i}: quantised bytes")
                wpt = wp[t0:t0 + valid].transpose(1, 0, 2)
                np.testing.assert_array_equal(prm[..., 1].view(np.uint32), np.ascontiguousarray(wpt[..., 1]).view(np.uint32),
                                              err_msg=f"{tag} {name} span {i}: scale bit patterns")
                # zero-points are integers: compared by VALUE.  (A head whose minimum is 0 gets zero = 0 - 0 / scale: the product
                # stores +0.0, the reference's code as hipcc compiles it -0.0 -- the one bit that differs; (q - zero) is the same.)
                np.testing.assert_array_equal(prm[..., 0], wpt[..., 0], err_msg=f"{tag} {name} span {i}: zero-points")

    # ---- prefill writer: dihip_kv_context_copy over interleaved qkv rows
    stride = (n + 2 * g) * H
    rows = torch.zeros(L, stride)
    rows[:, n * H:(n + g) * H] = heads.reshape(L, g * H)
    rows[:, (n + g) * H:] = vheads.reshape(L, g * H)
    rows_d = rows.to(dt).cuda()
    pool = ops.SpanPool(16, g, S, H, mode, dt)
    kv = ops.KVCacheSet(pool, 1, 4)
    kv.ensure(0, L)
    kv.sync()
    ops.kv_context_copy(kv.k_ptrs[0], rows_d[:, n * H:], stride, L, 0, g, H, S, mode)
    ops.kv_context_copy(kv.v_ptrs[0], rows_d[:, (n + g) * H:], stride, L, 0, g, H, S, mode)
    torch.cuda.synchronize()
    check(pool, kv, L, "context copy")
    # ---- decode writer: dihip_kv_append, token by token into a fresh cache
    pool2 = ops.SpanPool(16, g, S, H, mode, dt)
    kv2 = ops.KVCacheSet(pool2, 1, 4)
    kv2.ensure(0, L)
    kv2.sync()
    q_out = torch.empty(1, n * H, dtype=dt, device="cuda")
    old = torch.zeros(1, dtype=torch.int32, device="cuda")
    T = 40   # (covers the adversarial rows and a span boundary)
    for t in range(T):
        ops.kv_append(kv2, q_out, rows_d[t:t + 1].contiguous(), old, n, g, H)
        old += 1
    torch.cuda.synchronize()
    check(pool2, kv2, T, "decode append")


@pytest.mark.parametrize("mode", ["u4", "i8"])
def test_what_an_approximate_division_can_move(pkg, mode, capsys):
    """the reference's code with a <= 2-ulp division (a * v_rcp_f32(b)) against the same code with IEEE division: bounded, small"""
    exact, approx = _ref("ieee"), _ref("rcp")
    assert approx.ref_codec_division() != b"ieee"
    gen = torch.Generator().manual_seed(17)
    R = 1 << 15
    x = (torch.randn(R, 128, generator=gen) * torch.tensor([0.02, 1.0, 40.0])[torch.randint(0, 3, (R, 1), generator=gen)]).bfloat16().float().cuda()
    qe, pe = _ref_quantize(exact, mode, x)
    qa, pa = _ref_quantize(approx, mode, x)
    pe_i, pa_i = pe.cpu().numpy().view(np.int32).astype(np.int64), pa.cpu().numpy().view(np.int32).astype(np.int64)
    scale_ulps = np.abs(pe_i[:, 1] - pa_i[:, 1]).max()
    zero_moved = float((pe.cpu()[:, 0] != pa.cpu()[:, 0]).float().mean())
    zero_step = float((pe.cpu()[:, 0] - pa.cpu()[:, 0]).abs().max())
    if mode == "u4":
        ce = torch.stack([qe & 0xF, qe >> 4], -1).reshape(R, 128).short().cpu()
        ca = torch.stack([qa & 0xF, qa >> 4], -1).reshape(R, 128).short().cpu()
    else:
        ce, ca = qe.view(torch.int8).short().cpu(), qa.view(torch.int8).short().cpu()
    moved = float((ce != ca).float().mean())
    step = int((ce - ca).abs().max())
    with capsys.disabled():
        print(f"\n[kv codec, {mode}] reference code, a*rcp(b) vs IEEE division over {R} heads: scale differs by <= {scale_ulps} ulp, "
              f"zero differs on {zero_moved:.2%} of the heads (by <= {zero_step:g}), codes differ on {moved:.3%} of the elements (by <= {step})")
    assert scale_ulps <= 2 and zero_step <= 1.0 and step <= 2   # (a zero-point that moves by 1 moves every code of its head by 1)
    assert moved < 0.02, "an approximate division moves only elements that sit within ~1e-7 of a rounding boundary"
