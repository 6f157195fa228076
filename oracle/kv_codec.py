"""KV-cache span codec oracle in numpy (test infrastructure only).  PARITY UNPINNED:
the reference has no result test for int8/uint4 KV; restated from source.

Quantiser (one (zero, scale) pair per token-head, f32):
  span-attention/src/cache_quant/impl_i8.cuh:53-66 (Quant), :116-142 (Builder)
  span-attention/src/cache_quant/impl_u4.cuh:79-103, :157-184
  rounding = rintf (CONFIG_CACHE_ROUND_RNI, cache_quant/config.cuh:13); the reference's
  __fdividef (utils.cuh:29-32) is restated as IEEE division.
Span layout (csrc/core/kernel/cuda/cache/decoder_cache_append.cuh:33-87,
csrc/runtime/cache/virtual_cache.cpp:202-232):
  bytes [0, g*S*H')          data  [g][S][H']   H' = H*bits/8 ; u4: lo nibble = even d
  bytes [g*S*H', +g*S*8)     params[g][S] { f32 zero; f32 scale }   (quantised modes only)
"""
import numpy as np

from .numerics import bf16_bits, bf16_from_bits, ft_round

QRANGE = {"i8": (-128.0, 127.0), "u4": (0.0, 15.0)}


def span_bytes(g, S, H, mode, ft="bf16"):
    if mode == "none":
        return g * S * H * (4 if ft in ("f32", "fp32") else 2)
    data = g * S * H if mode == "i8" else g * S * H // 2
    return data + 2 * S * g * 4


def quant_params(x, mode):
    """x [..., H] f32 (FT-valued) -> (zero, scale) f32 [...]."""
    f32 = np.float32
    qmin, qmax = QRANGE[mode]
    x = np.asarray(x, f32)
    mx = x.max(axis=-1)
    mn = x.min(axis=-1)
    qs = ((mx - mn) / f32(qmax - qmin)).astype(f32)
    qs = np.maximum(qs, f32(1e-5))
    qz = (f32(qmin) - (mn / qs).astype(f32)).astype(f32)
    qz = np.minimum(qz, f32(qmax))
    if mode == "i8":
        qz = np.maximum(qz, f32(qmin))
    qz = np.rint(qz).astype(f32)
    return qz, qs


def quantize(x, zero, scale, mode):
    f32 = np.float32
    qmin, qmax = QRANGE[mode]
    t = (zero[..., None] + (np.asarray(x, f32) / scale[..., None]).astype(f32)).astype(f32)
    t = np.minimum(t, f32(qmax))
    if mode == "i8":
        t = np.maximum(t, f32(qmin))
    t = np.rint(t)
    if mode == "i8":
        return t.astype(np.int8)
    # static_cast<uint32_t>(tmp) & 0xf (impl_u4.cuh:79-93, :27-29): the device float -> u32 convert
    # SATURATES, so a negative tmp (reachable: zero is clamped at 15 with no lower clamp on the
    # elements, e.g. an all-negative head) becomes 0 -- it does not wrap mod 16.
    return (np.maximum(t, 0).astype(np.int64) & 0xF).astype(np.uint8)


def dequantize(qv, zero, scale):
    return ((qv.astype(np.float32) - zero[..., None]).astype(np.float32) * scale[..., None]).astype(np.float32)


class SpanCache:
    """Host model of one request's paged K or V cache: a list of span byte buffers carved
    from a pool with a non-contiguous (strided) allocation order, as the reference's span
    attention test does (span-attention/test/test_lib/test_quant_none.cpp:479-506)."""

    def __init__(self, g, S, H, mode="none", ft="bf16"):
        self.g, self.S, self.H, self.mode, self.ft = g, S, H, mode, ft
        self.nbytes = span_bytes(g, S, H, mode, ft)
        self.spans = []
        self.len = 0

    def _hb(self):
        return {"none": self.H * (4 if self.ft in ("f32", "fp32") else 2), "i8": self.H, "u4": self.H // 2}[self.mode]

    def ensure(self, ntokens):
        while len(self.spans) * self.S < ntokens:
            self.spans.append(np.zeros(self.nbytes, np.uint8))

    def write(self, pos, x):
        """x [g, H] f32 for token ``pos``."""
        g, S, H = self.g, self.S, self.H
        self.ensure(pos + 1)
        span = self.spans[pos // S]
        p = pos % S
        hb = self._hb()
        xr = ft_round(x, self.ft)
        data = span[: g * S * hb].reshape(g, S, hb)
        if self.mode == "none":
            if self.ft in ("f32", "fp32"):
                data[:, p, :] = xr.astype(np.float32).view(np.uint8).reshape(g, hb)
            elif self.ft in ("bf16",):
                data[:, p, :] = bf16_bits(xr).view(np.uint8).reshape(g, hb)
            else:
                data[:, p, :] = xr.astype(np.float16).view(np.uint8).reshape(g, hb)
        else:
            zero, scale = quant_params(xr, self.mode)
            qv = quantize(xr, zero, scale, self.mode)
            if self.mode == "i8":
                data[:, p, :] = qv.view(np.uint8)
            else:
                data[:, p, :] = (qv[:, 0::2] & 0xF) | ((qv[:, 1::2] & 0xF) << 4)
            params = span[g * S * hb:].view(np.float32).reshape(g, S, 2)
            params[:, p, 0] = zero
            params[:, p, 1] = scale
        self.len = max(self.len, pos + 1)

    def read(self, pos):
        """-> [g, H] f32 (dequantised)."""
        g, S, H = self.g, self.S, self.H
        span = self.spans[pos // S]
        p = pos % S
        hb = self._hb()
        data = span[: g * S * hb].reshape(g, S, hb)[:, p, :]
        if self.mode == "none":
            if self.ft in ("f32", "fp32"):
                return np.ascontiguousarray(data).view(np.float32).reshape(g, H).copy()
            if self.ft == "bf16":
                return bf16_from_bits(np.ascontiguousarray(data).view(np.uint16)).reshape(g, H)
            return np.ascontiguousarray(data).view(np.float16).astype(np.float32).reshape(g, H)
        params = span[g * S * hb:].view(np.float32).reshape(g, S, 2)[:, p, :]
        if self.mode == "i8":
            qv = np.ascontiguousarray(data).view(np.int8)
        else:
            qv = np.empty((g, H), np.uint8)
            qv[:, 0::2] = data & 0xF
            qv[:, 1::2] = data >> 4
        return dequantize(qv, params[:, 0], params[:, 1])

    def read_all(self, length=None):
        length = self.len if length is None else length
        return np.stack([self.read(t) for t in range(length)], 0) if length else np.zeros((0, self.g, self.H), np.float32)
