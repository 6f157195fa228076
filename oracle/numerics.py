"""Floating-point format helpers for the oracle (test infrastructure only).

bf16 rounding follows the reference's host bf16 type (round-to-nearest-even on
the upper 16 bits, NaN quieted): csrc/common/bfloat16_impl.hpp
(float -> bfloat16 conversion) and is identical to torch's ``.to(bfloat16)``.
"""
import numpy as np


def bf16_round(x):
    """Round an array to bfloat16 (RNE) and return it as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    nan = np.isnan(x)
    lsb = (u >> 16) & 1
    r = ((u + 0x7FFF + lsb) >> 16) << 16
    r = r.astype(np.uint32)
    out = r.view(np.float32).copy()
    out[nan] = np.nan
    return out.reshape(x.shape)


def bf16_bits(x):
    """float array -> uint16 bfloat16 bit patterns (RNE)."""
    return (bf16_round(x).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(b):
    """uint16 bfloat16 bit patterns -> float32."""
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32)


def f16_round(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def ft_round(x, ft):
    """Round to the 'FT' activation type of the reference (bf16 / f16 / f32)."""
    if ft in ("bf16", "bfloat16"):
        return bf16_round(x)
    if ft in ("f16", "fp16", "float16"):
        return f16_round(x)
    if ft in ("f32", "fp32", "float32"):
        return np.asarray(x, dtype=np.float32)
    raise ValueError(f"unknown FT {ft}")


def check_equal(ref, out):
    """The reference's comparison metric (tests/cpp/test_common.h.in:82-130):
    max over elements of min(|a-b|, |a-b|/|b|) with a=ref, b=out."""
    a = np.asarray(ref, dtype=np.float64).ravel()
    b = np.asarray(out, dtype=np.float64).ravel()
    d = np.abs(a - b)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(d / b)
    rel = np.where(np.isnan(rel), np.inf, rel)
    eps = np.minimum(d, rel)
    eps = np.where(np.isnan(eps), np.inf, eps)
    return float(eps.max()) if eps.size else 0.0
