"""Weight-only quantiser oracle (test infrastructure only).

Restates python/pyhie/allspark/model/quantization_utils.py of the reference:
  * quantize_gemm_weight_a16w8_torch   :158-217  -> iq_quantize_a16w8
  * quantize_gemm_weight_a16w4_torch   :240-304  -> iq_quantize_a16w4
  * depack_gptq_weight / depack_gptq_zero :331-351, repack_gptq_to_a16wX :391-437
and the 4-bit pack convention of
csrc/core/operator/general/gemm_lowp/gemm_a16w4.h:25-33 (byte = lo nibble even
n | hi nibble odd n; KATs 250 <-> [10, 15], 67 <-> [3, 4]).

All arithmetic is IEEE float32 in the order the torch code performs it, so the
results are byte-exact against the reference quantiser (checked by
tests/test_oracle_quant.py against tests/golden/quantizer_*.npz).
"""
import numpy as np

from .numerics import ft_round


def _pad_k(fdata, group):
    K, N = fdata.shape
    kstride = (K + group - 1) // group * group
    if kstride != K:
        pad = np.repeat(fdata[-1:, :], kstride - K, axis=0)  # repeat last row (:189-191)
        fdata = np.concatenate([fdata, pad], axis=0)
    return fdata


def _iq_params(data, qmin, qmax):
    """data: [N, G, group] float32 (values already FT-representable)."""
    f32 = np.float32
    fmax = data.max(axis=-1, keepdims=True).astype(f32)
    fmin = data.min(axis=-1, keepdims=True).astype(f32)
    scale = ((fmax - fmin) / f32(qmax - qmin)).astype(f32)       # :199
    scale = np.where(scale == 0, f32(1), scale).astype(f32)      # :200-202
    zero = (f32(qmin) - (fmin / scale).astype(f32)).astype(f32)  # :203 (NOT rounded)
    res = ((data / scale).astype(f32) + zero).astype(f32)        # :205
    q = np.rint(np.clip(res, f32(qmin), f32(qmax)))              # torch.round = half-to-even
    return q, scale, zero


def iq_quantize_a16w8(fdata, group_size=-1, ft="bf16"):
    """InstantQuant int8.  fdata [K, N] holding FT-representable values.
    Returns (q int8 [K,N], scales f32-valued-FT [G,N], zeros f32-valued-FT [G,N])."""
    fdata = ft_round(fdata, ft)
    K, N = fdata.shape
    group = K if group_size in (-1, None) else int(group_size)
    padded = _pad_k(fdata, group)
    data = padded.T.reshape(N, -1, group)
    q, scale, zero = _iq_params(data, -128.0, 127.0)
    q = q.reshape(N, -1).T.astype(np.int8)[:K, :]
    scale = ft_round(scale.reshape(N, -1).T, ft)
    zero = ft_round(zero.reshape(N, -1).T, ft)
    return np.ascontiguousarray(q), np.ascontiguousarray(scale), np.ascontiguousarray(zero)


def pack_u4(q):
    """[K, N] uint8 values 0..15 -> [K, ceil(N/2)] bytes, lo nibble = even n.
    (quantization_utils.py:297 ``(q[:,1::2] << 4) | (q[:,0::2] & 0xf)``;
    tests/cpp/operator/cuda/operator_gemm_lowp_test.cpp:17-29 PackU8ToU4x2.)"""
    q = np.asarray(q, dtype=np.uint8)
    K, N = q.shape
    if N % 2:
        q = np.concatenate([q, np.zeros((K, 1), np.uint8)], axis=1)
    return ((q[:, 1::2] << 4) | (q[:, 0::2] & 0xF)).astype(np.uint8)


def unpack_u4(packed, N):
    """Inverse of pack_u4 (csrc/core/kernel/cuda/gemm_lowp/convert_4bit.h:9-16)."""
    packed = np.asarray(packed, dtype=np.uint8)
    K = packed.shape[0]
    out = np.empty((K, packed.shape[1] * 2), np.uint8)
    out[:, 0::2] = packed & 0xF
    out[:, 1::2] = packed >> 4
    return out[:, :N]


def iq_quantize_a16w4(fdata, group_size=-1, ft="bf16"):
    """InstantQuant uint4.  Returns (packed u8 [K, ceil(N/2)], scales [G,N], zeros [G,N])."""
    fdata = ft_round(fdata, ft)
    K, N = fdata.shape
    group = K if group_size in (-1, None) else int(group_size)
    padded = _pad_k(fdata, group)
    nstride = (N + 1) // 2 * 2
    if nstride != N:  # ConstantPad2d with 0 (:271-275)
        padded = np.concatenate([padded, np.zeros((padded.shape[0], nstride - N), np.float32)], axis=1)
    data = padded.T.reshape(nstride, -1, group)
    q, scale, zero = _iq_params(data, 0.0, 15.0)
    q = q.reshape(nstride, -1).T.astype(np.uint8)
    packed = pack_u4(q)[:K, :]
    scale = ft_round(scale.reshape(nstride, -1).T[:, :N], ft)
    zero = ft_round(zero.reshape(nstride, -1).T[:, :N], ft)
    return np.ascontiguousarray(packed), np.ascontiguousarray(scale), np.ascontiguousarray(zero)


# ---------------------------------------------------------------- GPTQ ----
def depack_gptq_weight(qweight, bits=4):
    """qweight int32 [K*bits/32, N] -> [K, N] (quantization_utils.py:331-340):
    row r of qweight holds 32/bits consecutive k, lowest bits first."""
    qweight = np.asarray(qweight).astype(np.int64) & 0xFFFFFFFF
    per = 32 // bits
    shifts = (np.arange(per, dtype=np.int64) * bits)[None, :, None]
    w = (qweight[:, None, :] >> shifts) & ((1 << bits) - 1)
    return w.reshape(-1, qweight.shape[-1]).astype(np.int16 if bits == 8 else np.int8)


def depack_gptq_zero(qzeros, bits=4):
    """qzeros int32 [G, N*bits/32] -> [G, N], +1 (quantization_utils.py:343-351)."""
    qzeros = np.asarray(qzeros).astype(np.int64) & 0xFFFFFFFF
    per = 32 // bits
    shifts = (np.arange(per, dtype=np.int64) * bits)[None, None, :]
    z = (qzeros[:, :, None] >> shifts) & ((1 << bits) - 1)
    z = z + 1
    return z.reshape(qzeros.shape[0], -1).astype(np.int16 if bits == 8 else np.int8)


def repack_gptq_to_a16wx(qweight, qzeros, scales, bits, ft="bf16"):
    """AutoGPTQ checkpoint tensors -> allspark (weight, scales, zeros)
    (quantization_utils.py:391-437).  bits==4: packed u8 [K, N/2]; bits==8: int8 [K,N]
    (forced cast, :421-424)."""
    q = depack_gptq_weight(qweight, bits).reshape(-1, np.asarray(qweight).shape[-1])
    if bits == 4:
        assert (q >= 0).all()
        w = pack_u4(q.astype(np.uint8))
    elif bits == 8:
        w = q.astype(np.int8)
    else:
        raise ValueError(f"not supported quant_bits: {bits}")
    if qzeros is not None:
        z = ft_round(depack_gptq_zero(qzeros, bits).astype(np.float32), ft)
    else:
        z = np.zeros_like(np.asarray(scales, dtype=np.float32))
    return w, ft_round(scales, ft), z


# ------------------------------------------------- reference test quantiser
def test_quant_weight(fdata, group_size, qmin, qmax, ft="bf16"):
    """The *test-side* quantiser of the reference
    (tests/cpp/operator/cuda/operator_gemm_lowp_test.cpp:31-136,
    ComputeQuantParam + CPU_Quant_Weight_PerC/SubC): like IQ but the zero point
    is clamped to [qmin, qmax] and K is not padded."""
    f32 = np.float32
    fdata = ft_round(fdata, ft)
    K, N = fdata.shape
    group = K if group_size in (-1, None) else int(group_size)
    G = (K + group - 1) // group
    q = np.empty((K, N), np.float32)
    scales = np.empty((G, N), np.float32)
    zeros = np.empty((G, N), np.float32)
    for g in range(G):
        blk = fdata[g * group:(g + 1) * group]
        fmax = blk.max(axis=0).astype(f32)
        fmin = blk.min(axis=0).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            s = ((fmax - fmin) / f32(qmax - qmin)).astype(f32)
            z = (f32(qmin) - (fmin / s).astype(f32)).astype(f32)
            z = np.maximum(f32(qmin), np.minimum(f32(qmax), z))
            v = ((blk / s).astype(f32) + z).astype(f32)
        q[g * group:(g + 1) * group] = np.rint(np.maximum(f32(qmin), np.minimum(f32(qmax), v)))
        scales[g], zeros[g] = s, z
    return q, ft_round(scales, ft), ft_round(zeros, ft)
