"""Glue-op oracle: RMSNorm, RoPE, SwiGLU, greedy sampling (test infrastructure only).

RMSNorm  = LayerNormNoBeta: csrc/core/kernel/cpu/layernorm.cpp:110-157 (layerNormNobeta):
           rstd = 1/sqrt(mean(x^2) + eps); out = (gamma * x) * rstd   (that operation order).
RoPE     = csrc/core/kernel/cpu/rotary.cpp:22-106, rotate-half (NeoX) convention on the q and k
           heads of the fused qkv row: out[d] = x[d] cos - x[d+H/2] sin ;
           out[d+H/2] = x[d+H/2] cos + x[d] sin, angle = pos * inv_freq[d],
           inv_freq[d] = base^(-2d/H) (python/pyhie/allspark/model/model_base.py rotary inv_freq).
SwiGLU   = SiLU-activated gate GEMM x up GEMM (python/pyhie/allspark/model/qwen_v15.py:314-335).
greedy   = GenerateOp with top_k = 1 (csrc/core/operator/generate_opt/generate/generate_op.cpp).
"""
import numpy as np


def rmsnorm(x, gamma, eps=1e-6):
    x = np.asarray(x, np.float32)
    gamma = np.asarray(gamma, np.float32)
    var = np.einsum("...k,...k->...", x, x, dtype=np.float64)[..., None] / x.shape[-1]   # sum of squares in f64
    rstd = (1.0 / np.sqrt(var.astype(np.float32) + np.float32(eps))).astype(np.float32)
    return ((gamma * x).astype(np.float32) * rstd).astype(np.float32)


def rope_inv_freq(H, base=1000000.0):
    return (1.0 / (base ** (np.arange(0, H, 2, dtype=np.float64) / H))).astype(np.float32)


def rope(x, pos, inv_freq):
    """x [..., heads, H] f32; pos scalar or [...] broadcastable. rotate-half convention."""
    x = np.asarray(x, np.float32)
    H = x.shape[-1]
    half = H // 2
    ang = (np.asarray(pos, np.float32)[..., None, None] * inv_freq[None, :]).astype(np.float32)
    cos, sin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    x1, x2 = x[..., :half], x[..., half:]
    out = np.empty_like(x)
    out[..., :half] = x1 * cos - x2 * sin
    out[..., half:] = x2 * cos + x1 * sin
    return out


def silu(x, dtype=np.float64):
    """dtype: the type exp / the division run in (float32 for the full-depth comparisons: 40M elements per layer)."""
    x = np.asarray(x, np.float32)
    xd = x.astype(dtype, copy=False)
    return (xd / (dtype(1.0) + np.exp(-xd))).astype(np.float32, copy=False)


def greedy(logits):
    return np.argmax(np.asarray(logits), axis=-1)
