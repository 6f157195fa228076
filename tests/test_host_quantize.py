"""Host logic: dash-infer_amd/quantize.py (torch) reproduces the reference quantiser byte-exactly
(golden vectors generated from python/pyhie/allspark/model/quantization_utils.py).  CPU only."""
import os
import re

import numpy as np
import torch


def test_torch_quantiser_byte_exact_vs_reference_goldens(pkg, golden_dir):
    from dash_infer_amd import quantize
    g = np.load(os.path.join(golden_dir, "quantizer_iq.npz"))
    tags = sorted({re.match(r"w[48]_(.*)_[wqsz]$", k).group(1) for k in g.files})
    n = 0
    for tag in tags:
        ft, K, N, G = re.match(r"(bf16|f16)_K(\d+)_N(\d+)_G(-?\d+)$", tag).groups()
        dt = torch.bfloat16 if ft == "bf16" else torch.float16
        for wb in (8, 4):
            w = torch.from_numpy(g[f"w{wb}_{tag}_w"]).to(dt)
            q, s, z = quantize.quantize(w, wb, int(G))
            np.testing.assert_array_equal(q.numpy(), g[f"w{wb}_{tag}_q"], err_msg=f"{wb} {tag}")
            np.testing.assert_array_equal(s.float().numpy(), g[f"w{wb}_{tag}_s"], err_msg=f"{wb} {tag}")
            np.testing.assert_array_equal(z.float().numpy(), g[f"w{wb}_{tag}_z"], err_msg=f"{wb} {tag}")
            n += 1
    assert n >= 20


def test_gptq_like_zeros_are_integers():
    from dash_infer_amd import quantize
    w = (torch.randn(256, 32) * 0.02).to(torch.bfloat16)
    q, s, z = quantize.quantize(w, 4, 128, gptq_like_zeros=True)
    zf = z.float()
    assert torch.equal(zf, zf.round()) and zf.min() >= 1 and zf.max() <= 16
