"""ctypes bindings to the oracle's C libraries (test infrastructure only).

  oracle/_ref/liboracle_c.so     - our plain-C restatement (oracle/c/oracle_c.c)
  oracle/_ref/libdashinfer_ref.so - the reference's own host loops compiled from
                                    /root/reference by oracle/Makefile (optional)
  oracle/_ref/libdashinfer_ref_attn.so - the reference's x86 decoder attention
                                    (cpu_dec_single_mqa + kernel/cpu/mha.cpp), same recipe
Both are built by ``make -C oracle`` (also run by __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REFDIR = os.path.join(_HERE, "_ref")
FT_CODE = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "f16": 2, "fp16": 2, "float16": 2}
KV_CODE = {"none": 0, "i8": 1, "u4": 2}
ACT_CODE = {None: 0, "none": 0, "tanh": 1, "gelu_erf": 2, "gelu_tanh": 3, "relu": 4, "silu": 5, "sigmoid": 6}

_fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_REFDIR, "liboracle_c.so")
    if force or not os.path.exists(so):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return so


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_span_bytes.restype = C.c_size_t
        _lib.orc_round_ft.restype = C.c_float
    return _lib


def reflib():
    """The compiled reference loops, or None when oracle/_ref/libdashinfer_ref.so is absent."""
    global _ref
    if _ref is None:
        p = os.path.join(_REFDIR, "libdashinfer_ref.so")
        if not os.path.exists(p):
            try:
                subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)
            except Exception:
                pass
        _ref = C.CDLL(p) if os.path.exists(p) else False
    return _ref or None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def gemm_a16wx(A, B, S, Z, group, wbits, alpha=1.0, bias=None, act=None, ft="bf16", x86bf16=False):
    """Sequential-f32 CPU_SubC_Ref semantics (+bias/activation); returns float32 [M,N]."""
    A, S, Z = _f32(A), _f32(S), _f32(Z)
    B = np.ascontiguousarray(B)
    M, K = A.shape
    N = S.shape[-1]
    Cout = np.empty((M, N), np.float32)
    b = _f32(bias) if bias is not None else None
    if x86bf16:
        lib().orc_gemm_a16wx_x86bf16(_ptr(A), _ptr(B), _ptr(S), _ptr(Z), _ptr(b), _ptr(Cout), M, N, K,
                                     int(group), int(wbits), C.c_float(alpha), ACT_CODE[act])
    else:
        lib().orc_gemm_a16wx(_ptr(A), _ptr(B), _ptr(S), _ptr(Z), _ptr(b), _ptr(Cout), M, N, K,
                             int(group), int(wbits), C.c_float(alpha), ACT_CODE[act], FT_CODE[ft])
    return Cout


def span_bytes(g, S, H, mode, ft):
    return int(lib().orc_span_bytes(g, S, H, KV_CODE[mode], FT_CODE[ft]))


def span_write_head(span, x, head, pos, g, S, H, mode, ft):
    x = _f32(x)
    lib().orc_span_write_head(_ptr(span), _ptr(x), head, pos, g, S, H, KV_CODE[mode], FT_CODE[ft])


def span_read_head(span, head, pos, g, S, H, mode, ft):
    x = np.empty(H, np.float32)
    lib().orc_span_read_head(_ptr(span), _ptr(x), head, pos, g, S, H, KV_CODE[mode], FT_CODE[ft])
    return x


def span_attn_decode(q, kspans, vspans, length, n, g, H, S, mode, ft, alpha):
    """q [n,H] f32; kspans/vspans: lists of uint8 numpy span buffers. Returns [n,H] f32."""
    q = _f32(q)
    out = np.empty((n, H), np.float32)
    ka = (C.c_void_p * len(kspans))(*[s.ctypes.data for s in kspans])
    va = (C.c_void_p * len(vspans))(*[s.ctypes.data for s in vspans])
    lib().orc_span_attn_decode(_ptr(out), _ptr(q), ka, va, int(length), n, g, H, S, KV_CODE[mode],
                               FT_CODE[ft], C.c_float(alpha))
    return out


def prefill_attn(q, k, v, n, g, H, alpha, causal=True):
    q, k, v = _f32(q), _f32(k), _f32(v)
    Lq, Lk = q.shape[0], k.shape[0]
    out = np.empty((Lq, n, H), np.float32)
    lib().orc_prefill_attn(_ptr(out), _ptr(q), _ptr(k), _ptr(v), Lq, Lk, n, g, H, C.c_float(alpha),
                           1 if causal else 0)
    return out


# ------------------------------------------------------- compiled reference loops (optional)
def ref_gemm_a16w8(A, B, S, Z, group, alpha, ft):
    r = reflib()
    A, S, Z = _f32(A), _f32(S), _f32(Z)
    B = np.ascontiguousarray(B, dtype=np.int8)
    M, K = A.shape
    N = B.shape[1]
    out = np.empty((M, N), np.float32)
    r.ref_gemm_a16w8(_ptr(A), _ptr(B), _ptr(S), _ptr(Z), _ptr(out), M, N, K, int(group),
                     C.c_float(alpha), 0 if FT_CODE[ft] == 0 else 1)
    return out


def ref_gemm_a16w4_unpacked(A, Bu8, S, Z, group, alpha, ft):
    r = reflib()
    A, S, Z = _f32(A), _f32(S), _f32(Z)
    Bu8 = np.ascontiguousarray(Bu8, dtype=np.uint8)
    M, K = A.shape
    N = Bu8.shape[1]
    out = np.empty((M, N), np.float32)
    r.ref_gemm_a16w4_unpacked(_ptr(A), _ptr(Bu8), _ptr(S), _ptr(Z), _ptr(out), M, N, K, int(group),
                              C.c_float(alpha), 0 if FT_CODE[ft] == 0 else 1)
    return out


def ref_gemm_a16w4_perc_packed(A, Bpack, S, Z, N):
    r = reflib()
    A = _f32(A)
    S = _f32(np.concatenate([np.ravel(S), [0.0]]))
    Z = _f32(np.concatenate([np.ravel(Z), [0.0]]))
    Bpack = np.ascontiguousarray(Bpack, dtype=np.uint8)
    M, K = A.shape
    out = np.empty((M, N), np.float32)
    r.ref_gemm_a16w4_perc_packed(_ptr(A), _ptr(Bpack), _ptr(S), _ptr(Z), _ptr(out), M, N, K)
    return out


def ref_pack_u8_to_u4x2(data):
    r = reflib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    K, N = data.shape
    NP = (N + 1) // 2
    out = np.empty((K, NP), np.uint8)
    r.ref_pack_u8_to_u4x2(_ptr(data), _ptr(out), N, NP, K)
    return out


def ref_test_quant_weight(W, group, qbits):
    r = reflib()
    W = _f32(W)
    K, N = W.shape
    G = (K + group - 1) // group if group > 0 else 1
    Q = np.empty((K, N), np.float32)
    S = np.empty((G, N), np.float32)
    Z = np.empty((G, N), np.float32)
    r.ref_test_quant_weight(_ptr(W), _ptr(Q), _ptr(S), _ptr(Z), N, K, int(group), int(qbits))
    return Q, S, Z


def ref_prefill_check(concat, output, alpha, causal=True, feps=1e-3):
    """concat [batch, seqlen, 3, nhead, phead], output [batch, seqlen, nhead, phead]."""
    r = reflib()
    concat, output = _f32(concat), _f32(output)
    b, s, _, nh, ph = concat.shape
    return bool(r.ref_prefill_check(_ptr(concat), _ptr(output), b, s, nh, ph, C.c_float(alpha),
                                    1 if causal else 0, C.c_float(feps)))


# ---- the reference's own x86 decoder attention (oracle/ref_attn_shim.cpp -> _ref/libdashinfer_ref_attn.so) -----------------
_ref_attn = None


def ref_attn_lib():
    """BatchMQAOp's cpu_dec_single_mqa + the AVX2 softmax / batch helpers of kernel/cpu/mha.cpp, compiled from /root/reference
    by oracle/Makefile (needs AVX2 on the host), or None when absent."""
    global _ref_attn
    if _ref_attn is None:
        p = os.path.join(_REFDIR, "libdashinfer_ref_attn.so")
        ok = os.path.exists(p)
        if ok:
            try:
                ok = "avx2" in open("/proc/cpuinfo").read()
            except OSError:
                ok = False
        _ref_attn = C.CDLL(p) if ok else False
    return _ref_attn or None


def ref_decode_attention_step(qkv, k_cache, v_cache, step, n, g, H, alpha):
    """One decoder step of the reference's x86 attention for `batch` requests at the same step.  qkv f32 [batch, (n + 2 g) H]
    (this step's rotated q | k | v), k_cache / v_cache f32 [batch, cache_max_len, g H] holding step - 1 earlier tokens (this step's
    rows are appended in place at position step - 1).  -> out f32 [batch, n H]."""
    l = ref_attn_lib()
    qkv = _f32(qkv)
    batch = qkv.shape[0]
    assert k_cache.dtype == np.float32 and v_cache.dtype == np.float32 and k_cache.flags.c_contiguous and v_cache.flags.c_contiguous
    out = np.empty((batch, n * H), np.float32)
    l.ref_dec_single_mqa.argtypes = [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
    l.ref_dec_single_mqa(out, qkv, k_cache, v_cache, batch, int(step), k_cache.shape[1], n, H, g, float(alpha))
    return out


def ref_vsoftmax(row, temperature=1.0):
    l = ref_attn_lib()
    r = _f32(row).copy()
    l.ref_vsoftmax.argtypes = [_fp, C.c_int, C.c_float]
    l.ref_vsoftmax(r, r.shape[0], float(temperature))
    return r
