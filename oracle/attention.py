"""Attention oracle in numpy (test infrastructure only).

decode : cpu_dec_single_mqa, csrc/core/operator/generate_opt/batch_mqa/batch_mqa_op.cpp:140-179
         score = alpha * Q.K^T (cblas_sgemm alpha) -> f32 softmax (csrc/core/kernel/cpu/mha.cpp
         :783-806) -> P.V ; query head h uses KV head h // (n // g) (mha.cpp:748-765).
prefill: causal softmax(alpha Q K^T) V, host loop of
         tests/cpp/kernel/cuda/kernel_mhaprefill_test.cpp:119-250, generalised to GQA and
         a cached prefix (query i sees keys j <= i + Lk - Lq), which is what
         xformer_prefill_attention computes (csrc/core/kernel/cuda/xformer_mha/xformer_mha.h:26-41).
PARITY UNPINNED at the MKL cblas boundary (third_party/mkl_2022.0.2.tar.gz is an LFS stub):
the k-sum order of sgemm is not reproduced; sums here are float64.
"""
import numpy as np


def softmax_rows(s, dtype=np.float64):
    s = np.asarray(s, dtype)
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(axis=-1, keepdims=True)


def decode_attention(q, k, v, alpha):
    """q [n,H]; k,v [L,g,H] (already dequantised). -> [n,H] float32."""
    q = np.asarray(q, np.float64)
    k = np.asarray(k, np.float64)
    v = np.asarray(v, np.float64)
    n, H = q.shape
    g = k.shape[1]
    hpg = n // g
    out = np.empty((n, H), np.float64)
    for h in range(n):
        grp = h // hpg
        p = softmax_rows(alpha * (k[:, grp, :] @ q[h]))
        out[h] = p @ v[:, grp, :]
    return out.astype(np.float32)


def prefill_attention(q, k, v, alpha, causal=True, dtype=np.float64, threads=1):
    """q [Lq,n,H]; k,v [Lk,g,H] -> [Lq,n,H] float32.  dtype: accumulation type of the two contractions and the softmax
    (float64 by default; float32 = what cblas_sgemm + the f32 softmax of the x86 path carry, used by the full-depth
    comparisons where a float64 pass over 28 layers x 2k tokens would take minutes).  threads > 1: heads on a thread pool
    (BLAS held to one thread per call meanwhile): the same per-head arithmetic up to the summation order of BLAS's blocking."""
    q = np.asarray(q, dtype)
    k = np.asarray(k, dtype)
    v = np.asarray(v, dtype)
    Lq, n, H = q.shape
    Lk, g, _ = k.shape
    hpg = n // g
    off = Lk - Lq
    out = np.empty((Lq, n, H), dtype)
    mask = None
    if causal:
        mask = np.arange(Lk)[None, :] > (np.arange(Lq)[:, None] + off)
    def head(h):
        grp = h // hpg
        s = dtype(alpha) * (q[:, h, :] @ k[:, grp, :].T)
        if mask is not None:
            s[mask] = -np.inf
        out[:, h, :] = softmax_rows(s, dtype) @ v[:, grp, :]
    if threads > 1 and n > 1:
        from concurrent.futures import ThreadPoolExecutor
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:  # plain loop: concurrent multi-threaded BLAS calls would only fight over one pool
            threadpool_limits = None
        if threadpool_limits is not None:
            with threadpool_limits(limits=1, user_api="blas"), ThreadPoolExecutor(min(threads, n)) as ex:
                list(ex.map(head, range(n)))
            return out.astype(np.float32)
    for h in range(n):
        head(h)
    return out.astype(np.float32)
