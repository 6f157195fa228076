"""Test helpers: numpy specification of the dihip tile-major weight layout
and small conveniences.  The layout is OUR design (not the reference's), so its spec lives with
the tests; the pack kernels in csrc/gemm_lowp.hip must reproduce it byte-exactly."""
import numpy as np

from oracle.quant import unpack_u4


def roundup(v, m):
    return (v + m - 1) // m * m


def pack_tile_major(wq, N, wbits):
    """wq: int8 [K,N] (wbits 8) | packed u8 [K,ceil(N/2)] (wbits 4) | uint16 [K,N] bits (wbits 16).
    Returns the uint32 image [NTILES, KT, 64 lanes, 4 dwords]."""
    K = wq.shape[0]
    ktile = {4: 128, 8: 64, 16: 32}[wbits]
    Kp, Np = roundup(K, ktile), roundup(N, 16)
    KT, NT = Kp // ktile, Np // 16
    if wbits == 4:
        q = np.zeros((Kp, Np), np.uint32)
        q[:K, :N] = unpack_u4(wq, N)
    elif wbits == 8:
        q = np.zeros((Kp, Np), np.uint32)
        q[:K, :N] = (wq.astype(np.int32) + 128).astype(np.uint32)
        q[K:, :] = 0
        q[:, N:] = 0
    else:
        q = np.zeros((Kp, Np), np.uint32)
        q[:K, :N] = wq.astype(np.uint32)
    out = np.zeros((NT, KT, 64, 4), np.uint32)
    # index helpers
    lane = np.arange(64)
    ni, kb = lane & 15, lane >> 4
    for nt in range(NT):
        cols = nt * 16 + ni  # [64]
        for kt in range(KT):
            if wbits == 4:
                for ks in range(4):
                    d = np.zeros(64, np.uint32)
                    for j in range(8):
                        k = kt * 128 + ks * 32 + kb * 8 + j
                        d |= q[k, cols] << np.uint32(4 * (j >> 1) + 16 * (j & 1))
                    out[nt, kt, :, ks] = d
            elif wbits == 8:
                for dw in range(4):
                    ks, j0 = dw >> 1, (dw & 1) * 4
                    d = np.zeros(64, np.uint32)
                    for j in range(4):
                        k = kt * 64 + ks * 32 + kb * 8 + j0 + j
                        d |= q[k, cols] << np.uint32(8 * j)
                    out[nt, kt, :, dw] = d
            else:
                for dw in range(4):
                    k = kt * 32 + kb * 8 + dw * 2
                    out[nt, kt, :, dw] = q[k, cols] | (q[k + 1, cols] << np.uint32(16))
    return out


def pack_sz(scales_bits, zeros_bits, N, K, group):
    """uint16 bit patterns [G,N] -> uint32 [NTILES, Gp, 16] (column-tile major; lo = scale, hi = zero)."""
    Np = roundup(N, 16)
    Kp = roundup(K, 128)
    G = scales_bits.shape[0]
    Gp = max(G, (Kp + group - 1) // group) if group and group > 0 else 1
    out = np.zeros((Gp, Np), np.uint32)
    out[:G, :N] = scales_bits.astype(np.uint32) | (zeros_bits.astype(np.uint32) << 16)
    return np.ascontiguousarray(out.reshape(Gp, Np // 16, 16).transpose(1, 0, 2))
