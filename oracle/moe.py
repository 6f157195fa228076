"""Mixture-of-experts oracle in numpy (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path).

Restates the reference's MOE operator, csrc/core/operator/general/moe/moe_op.cpp:338-460 (CUDA only: "MOE Operator
does not support CPU", :464-468 -- there is no reference CPU path to compile, so this follows the kernels' math):
  * router: float softmax over the experts with `sum + 1e-12` (csrc/core/kernel/cuda/softmax_low_reduce.cu:11-41),
    top-k of the probabilities, NOT renormalised (TopKKernelLauncher, moe_op.cpp:359-365); descending order, the
    lower expert index first on ties (the heap order of topk.cu is not pinned by any reference test: parity
    unpinned for exact ties);
  * expert FFN: UnaryGLU(SILU) over [gate | up] (csrc/core/kernel/cuda/unary.cu:122-132), then the down projection;
  * combine: out[t] = sum_k score[t, k] * y[t, k] accumulated in float in rank order (finalize_new_kernel,
    csrc/core/kernel/cuda/moe.cu:386-418), cast to FT.
Experts here are weight-only quantised (A16W8 / A16W4, oracle.gemm_ref) -- the reference's are bf16 / A8W8.
Intermediate rounding: SiLU(gate) * up and the expert output are rounded to FT once each (the kernels fuse the GLU
into the gate/up GEMV); the reference rounds gate, up and SiLU(gate) separately -- within the FT tolerance of the tests.
"""
import numpy as np

from . import gemm_ref
from .glue import silu
from .numerics import ft_round


def route(logits, top_k):
    """logits [T, E] -> (scores f32 [T, k], experts i32 [T, k])."""
    x = np.asarray(logits, np.float32)
    mx = x.max(axis=1, keepdims=True)
    e = np.exp((x - mx).astype(np.float32)).astype(np.float32)
    p = (e / (e.sum(axis=1, keepdims=True, dtype=np.float32) + np.float32(1e-12))).astype(np.float32)
    T, E = p.shape
    scores = np.zeros((T, top_k), np.float32)
    experts = np.zeros((T, top_k), np.int32)
    for t in range(T):
        order = sorted(range(E), key=lambda i: (-p[t, i], i))[:top_k]
        experts[t] = order
        scores[t] = p[t, order]
    return scores, experts


def experts_ffn(x, experts, scores, gate, up, down, group, wbits, ft="bf16"):
    """x FT [T, hidden]; gate / up / down: lists of (q, scales, zeros) per expert.  Returns FT-rounded f32 [T, hidden]."""
    T, hidden = x.shape
    out = np.zeros((T, hidden), np.float32)
    for t in range(T):
        acc = np.zeros(hidden, np.float32)
        for k in range(experts.shape[1]):
            e = int(experts[t, k])
            if e < 0:
                continue
            xr = x[t:t + 1]
            g = gemm_ref.gemm_a16wx(xr, *gate[e], group, wbits, ft="f32")
            u = gemm_ref.gemm_a16wx(xr, *up[e], group, wbits, ft="f32")
            a = ft_round((silu(g) * u).astype(np.float32), ft)
            y = ft_round(gemm_ref.gemm_a16wx(a, *down[e], group, wbits, ft="f32"), ft)
            acc = (acc + np.float32(scores[t, k]) * y[0]).astype(np.float32)
        out[t] = acc
    return ft_round(out, ft)
