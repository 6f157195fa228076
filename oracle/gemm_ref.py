"""Weight-only GEMM oracle in numpy (test infrastructure only).

Math: tests/cpp/operator/cuda/operator_gemm_lowp_test.cpp:138-219
  C[m,n] = FT(alpha * sum_k A[m,k] * ((q[k,n] - Z[k/G,n]) * S[k/G,n]))
(CPU_SubC_Ref / CPU_PerC_Ref / CPU_FP16W4_PerC_Ref) plus the op-level epilogue of
GemmA16W8GPU::Forward (csrc/core/operator/general/gemm_lowp/gemm_a16w8_gpu.cpp:169-248:
bias add, UnaryType activation).

``dequant`` reproduces the per-element f32 expression ``(float(q) - float(z)) * float(s)``
bit-exactly; the k-sum here is done in float64 ("exact" mode) - the sequential-f32 sum
of the reference lives in oracle/c (orc_gemm_a16wx) and in oracle/_ref (the reference's own
loop).  ``x86_bf16`` mirrors GemmOpCPU under matmul_precision=medium_bf16
(csrc/core/operator/general/gemm/gemm_op_cpu.cpp:75-126): bf16(x) . bf16(W_deq) -> f32.
"""
import numpy as np

from .numerics import bf16_round, ft_round
from .quant import unpack_u4


def dequant(q, scales, zeros, group, wbits, N=None, threads=1):
    """q: int8 [K,N] (wbits 8) or packed u8 [K,ceil(N/2)] (wbits 4). Returns f32 [K,N].
    threads > 1: row blocks (whole groups) on a thread pool -- the same per-element expression, for the full-depth
    comparisons that dequantise 28 layers of 7B-width matrices (numpy releases the GIL inside these array passes)."""
    scales = np.asarray(scales, np.float32)
    zeros = np.asarray(zeros, np.float32)
    N = scales.shape[-1] if N is None else N
    K0 = q.shape[0]
    g0 = K0 if group in (-1, None, 0) else int(group)
    if threads > 1 and K0 >= 2 * threads and (g0 == K0 or K0 % g0 == 0):
        from concurrent.futures import ThreadPoolExecutor
        unit = 1 if g0 == K0 else g0
        per = -(-(K0 // unit) // threads) * unit
        per = max(unit, min(per, (max(1, (1 << 21) // max(N, 1)) // unit) * unit))  # blocks of <= 2M elements stay cache-sized
        out = np.empty((K0, N), np.float32)

        def job(k0):
            k1 = min(K0, k0 + per)
            if g0 == K0:
                out[k0:k1] = dequant(q[k0:k1], scales, zeros, -1, wbits, N)
            else:
                out[k0:k1] = dequant(q[k0:k1], scales[k0 // g0:k1 // g0], zeros[k0 // g0:k1 // g0], g0, wbits, N)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(job, range(0, K0, per)))
        return out
    if wbits == 4:
        q = unpack_u4(q, N)
    K = q.shape[0]
    g = K if group in (-1, None, 0) else int(group)
    if K % g == 0 and zeros.ndim == 2 and zeros.shape[0] == K // g:
        # the same f32 expression per element, broadcast per group instead of through [K, N] gathers of the parameters
        # (a 7B-width matrix is 68M elements: the full-depth comparisons dequantise 28 layers of them)
        w = q.astype(np.float32).reshape(K // g, g, q.shape[1])
        w -= zeros[:, None, :]
        w *= scales[:, None, :]
        return w.reshape(K, q.shape[1])
    idx = np.arange(K) // g
    return ((q.astype(np.float32) - zeros[idx]).astype(np.float32) * scales[idx]).astype(np.float32)


def activation(v, act):
    if act in (None, "none"):
        return v
    if act == "relu":
        return np.maximum(v, 0)
    if act == "tanh":
        return np.tanh(v)
    if act == "sigmoid":
        return 1.0 / (1.0 + np.exp(-v))
    if act == "silu":  # oneDNN eltwise_swish alpha=1 on x86 / hie SiLU functor on GPU
        return v / (1.0 + np.exp(-v))
    if act == "gelu_erf":
        from math import erf
        return 0.5 * v * (1.0 + np.vectorize(erf)(v * 0.7071067811865476))
    if act == "gelu_tanh":
        return 0.5 * v * (1.0 + np.tanh(0.7978845608 * (v + 0.044715 * v ** 3)))
    raise ValueError(act)


def gemm_a16wx(x, q, scales, zeros, group, wbits, alpha=1.0, bias=None, act=None, ft="bf16",
               mode="exact", round_out=True):
    """x [M,K] FT-valued. mode: 'exact' (f64 k-sum) | 'x86_bf16'. Returns f32 [M,N]."""
    x = np.asarray(x, np.float32)
    w = dequant(q, scales, zeros, group, wbits)
    if mode == "x86_bf16":
        acc = bf16_round(x).astype(np.float64) @ bf16_round(w).astype(np.float64)
    else:
        acc = x.astype(np.float64) @ w.astype(np.float64)
    v = alpha * acc
    if bias is not None:
        v = v + np.asarray(bias, np.float64)[None, :]
    v = activation(v, act)
    v = v.astype(np.float32)
    if round_out and mode != "x86_bf16":
        v = ft_round(v, ft)
    return v
