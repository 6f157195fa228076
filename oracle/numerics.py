"""Floating-point format helpers for the oracle (test infrastructure only).

bf16 rounding follows the reference's host bf16 type (round-to-nearest-even on
the upper 16 bits, NaN quieted): csrc/common/bfloat16_impl.hpp
(float -> bfloat16 conversion) and is identical to torch's ``.to(bfloat16)``.
"""
import numpy as np


def bf16_round(x, threads=1):
    """Round an array to bfloat16 (RNE) and return it as float32.  threads > 1: row blocks of a 2-D array on a thread pool."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if threads > 1 and x.ndim == 2 and x.shape[0] >= 2 * threads:
        from concurrent.futures import ThreadPoolExecutor
        out = np.empty_like(x)
        per = max(1, min(-(-x.shape[0] // threads), (1 << 21) // max(x.shape[1], 1)))  # cache-sized blocks

        def job(r0):
            out[r0:r0 + per] = bf16_round(x[r0:r0 + per])
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(job, range(0, x.shape[0], per)))
        return out
    u = x.view(np.uint32)
    # u + 0x7FFF + lsb in 32 bits: it can only wrap for bit patterns >= 0xFFFF8000, which are NaNs (patched below)
    r = (u >> 16) & np.uint32(1)
    r += np.uint32(0x7FFF)
    with np.errstate(over="ignore"):
        r += u
    r &= np.uint32(0xFFFF0000)
    out = r.view(np.float32)
    nan = np.isnan(x)
    if nan.any():
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
