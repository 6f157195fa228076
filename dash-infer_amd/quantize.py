"""InstantQuant weight quantiser (host/offline logic, torch; runs on CPU or GPU tensors).

Mirrors python/pyhie/allspark/model/quantization_utils.py of the reference
(quantize_gemm_weight_a16w8_torch :158-217, quantize_gemm_weight_a16w4_torch :240-304): same
operator interface (weight [K, N] in FT -> (q, scales, zeros)), same arithmetic order, so a model
quantised here loads bit-identically to one quantised by the reference converter.  Byte-exact
parity with the reference's own outputs is asserted by tests/test_host_quantize.py against
tests/golden/quantizer_iq.npz.
"""
import torch


def _pad_k(fdata, group):
    K, N = fdata.shape
    kstride = (K + group - 1) // group * group
    if kstride != K:
        fdata = torch.cat((fdata, fdata[-1:, :].repeat(kstride - K, 1)), 0)
    return fdata


def _quant(data, qmin, qmax):
    qmax_t = torch.tensor(float(qmax), dtype=torch.float32, device=data.device)
    qmin_t = torch.tensor(float(qmin), dtype=torch.float32, device=data.device)
    fmax = torch.amax(data, dim=-1, keepdim=True).to(torch.float32)
    fmin = torch.amin(data, dim=-1, keepdim=True).to(torch.float32)
    scale = (fmax - fmin) / (qmax_t - qmin_t)
    scale = torch.where(scale == 0, torch.ones_like(scale), scale)
    zero = qmin_t - fmin / scale
    q = torch.round(torch.clamp((data / scale + zero).float(), qmin_t, qmax_t))
    return q, scale, zero


def quantize_a16w8(fdata, group_size=-1):
    """-> (int8 [K,N], scales FT [G,N], zeros FT [G,N])."""
    ftype = fdata.dtype
    K, N = fdata.shape
    group = K if group_size in (-1, None, 0) else int(group_size)
    data = _pad_k(fdata, group).transpose(1, 0).reshape(N, -1, group)
    q, scale, zero = _quant(data, -128, 127)
    q = q.view(N, -1).transpose(1, 0).contiguous().to(torch.int8)[:K, :]
    scale = scale.view(N, -1).transpose(1, 0).contiguous().to(ftype)
    zero = zero.view(N, -1).transpose(1, 0).contiguous().to(ftype)
    return q.contiguous(), scale, zero


def quantize_a16w4(fdata, group_size=-1):
    """-> (packed uint8 [K, ceil(N/2)], scales FT [G,N], zeros FT [G,N]); lo nibble = even n."""
    ftype = fdata.dtype
    K, N = fdata.shape
    group = K if group_size in (-1, None, 0) else int(group_size)
    padded = _pad_k(fdata, group)
    nstride = (N + 1) // 2 * 2
    if nstride != N:
        padded = torch.nn.functional.pad(padded, (0, nstride - N, 0, 0))
    data = padded.transpose(1, 0).reshape(nstride, -1, group)
    q, scale, zero = _quant(data, 0, 15)
    q = q.view(nstride, -1).transpose(1, 0).contiguous().to(torch.uint8)
    packed = (q[:, 1::2] << 4) | (q[:, 0::2] & 0xF)
    scale = scale.view(nstride, -1).transpose(1, 0).contiguous()[:, :N].to(ftype)
    zero = zero.view(nstride, -1).transpose(1, 0).contiguous()[:, :N].to(ftype)
    return packed[:K, :].contiguous(), scale.contiguous(), zero.contiguous()


def quantize(fdata, wbits, group_size=-1, gptq_like_zeros=False):
    q, s, z = (quantize_a16w8 if wbits == 8 else quantize_a16w4)(fdata, group_size)
    if gptq_like_zeros:
        # mimic depack_gptq_zero (integer zero points, +1 convention => [1, 16])
        z = torch.clamp(torch.round(z.float()), 1, 16).to(z.dtype)
    return q, s, z


# ---------------------------------------------------------------------- GPTQ checkpoints ----
def depack_gptq_weight(qweight, bits=4):
    """AutoGPTQ qweight int32 [K*bits/32, N] -> integers [K, N]: row r holds 32/bits consecutive k,
    lowest bits first (quantization_utils.py:331-340)."""
    per = 32 // bits
    q = qweight.to(torch.int64) & 0xFFFFFFFF
    shifts = (torch.arange(per, dtype=torch.int64, device=q.device) * bits)[None, :, None]
    return ((q[:, None, :] >> shifts) & ((1 << bits) - 1)).reshape(-1, q.shape[-1])


def depack_gptq_zero(qzeros, bits=4):
    """AutoGPTQ qzeros int32 [G, N*bits/32] -> [G, N], stored minus one (quantization_utils.py:343-351)."""
    per = 32 // bits
    z = qzeros.to(torch.int64) & 0xFFFFFFFF
    shifts = (torch.arange(per, dtype=torch.int64, device=z.device) * bits)[None, None, :]
    return (((z[:, :, None] >> shifts) & ((1 << bits) - 1)) + 1).reshape(z.shape[0], -1)


def repack_gptq(qweight, qzeros, scales, bits, dtype=torch.bfloat16):
    """AutoGPTQ checkpoint tensors -> the (weight, scales, zeros) triple GemmA16W4 / GemmA16W8 take
    (quantization_utils.py:391-437): bits 4 -> u8 [K, ceil(N/2)] with the low nibble = even n; bits 8 ->
    int8 [K, N] (the reference's forced cast); zeros = stored zero + 1 in FT; scales in FT."""
    q = depack_gptq_weight(qweight, bits)
    if bits == 4:
        if q.shape[1] % 2:
            q = torch.nn.functional.pad(q, (0, 1))
        w = ((q[:, 0::2] & 0xF) | ((q[:, 1::2] & 0xF) << 4)).to(torch.uint8)
    elif bits == 8:
        w = q.to(torch.uint8).view(torch.int8) if q.dtype != torch.int8 else q
    else:
        raise ValueError(f"not supported quant_bits: {bits}")
    z = depack_gptq_zero(qzeros, bits).to(torch.float32).to(dtype) if qzeros is not None else torch.zeros_like(scales, dtype=dtype)
    return w.contiguous(), scales.to(dtype).contiguous(), z.contiguous()
