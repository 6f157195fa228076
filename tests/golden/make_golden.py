#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference); the GPU box uses the
committed .npz files.  What is executed is the reference's own Python
quantiser, python/pyhie/allspark/model/quantization_utils.py: the module cannot
be imported as a package (it pulls in the protobuf-generated model_base and the
compiled _allspark extension), so its source is read from /root/reference at
run time, the two package-relative import lines are dropped, and the module
body is executed in a namespace that supplies the few names it needs
(torch, numpy, re, a QuantizeConfig stand-in).  No reference source is copied
into this repository.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import re
import sys
import types

import numpy as np
import torch

REF = os.environ.get("DASHINFER_REFERENCE", "/root/reference")
QSRC = os.path.join(REF, "python/pyhie/allspark/model/quantization_utils.py")
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_quantizer():
    src = open(QSRC).read()
    src = src.replace("from .model_base import *", "")
    src = src.replace("from ..quantization import QuantizeConfig", "")

    class _QuantMode:
        A16W8, A16W4, A8W8, FP8A8W8 = "A16W8", "A16W4", "A8W8", "FP8A8W8"

    class QuantizeConfig:
        QuantMode = _QuantMode

    ns = {"re": re, "np": np, "torch": torch, "QuantizeConfig": QuantizeConfig,
          "make_tensor": lambda name: name, "__name__": "reference_quantization_utils"}
    exec(compile(src, QSRC, "exec"), ns)
    return types.SimpleNamespace(**ns)


def qcfg(weight_type, subchannel, group):
    c = types.SimpleNamespace()
    c.weight_type = weight_type
    c.extra_option = {"SubChannel": subchannel, "GroupSize": group}
    c.quantize_mode = None
    return c


def bits(t):
    """bf16/f16 torch tensor -> float32 numpy (exact)."""
    return t.to(torch.float32).numpy()


def main():
    ref = load_reference_quantizer()
    g = torch.Generator().manual_seed(20240925)
    cases = {}
    # (K, N, group) incl. ragged K (pad-by-repeat), odd N (4-bit pad), zero-range column
    shapes = [(256, 64, -1), (256, 64, 128), (200, 48, 64), (130, 34, 128), (384, 33, 128), (96, 16, 32)]
    for ft_name, ft in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        for (K, N, G) in shapes:
            w = (torch.randn(K, N, generator=g) * 0.05).to(ft)
            w[:, 3] = w[0, 3]  # a constant column -> scale==0 -> 1 branch
            sub = G != -1
            tag = f"{ft_name}_K{K}_N{N}_G{G}"
            if N % 2 == 0 or True:
                q8, s8, z8 = ref.quantize_gemm_weight_a16w8_torch(w.clone(), qcfg("int8", sub, G))
                cases[f"w8_{tag}_w"] = bits(w)
                cases[f"w8_{tag}_q"] = q8.numpy()
                cases[f"w8_{tag}_s"] = bits(s8)
                cases[f"w8_{tag}_z"] = bits(z8)
            q4, s4, z4 = ref.quantize_gemm_weight_a16w4_torch(w.clone(), qcfg("uint4", sub, G))
            cases[f"w4_{tag}_w"] = bits(w)
            cases[f"w4_{tag}_q"] = q4.numpy()
            cases[f"w4_{tag}_s"] = bits(s4)
            cases[f"w4_{tag}_z"] = bits(z4)
    np.savez_compressed(os.path.join(OUT, "quantizer_iq.npz"), **cases)

    # ---- GPTQ repack: synthetic AutoGPTQ tensors -> reference repack
    gcases = {}
    for bits_ in (4, 8):
        K, N, G = 256, 64, 128
        per = 32 // bits_
        qw = torch.randint(0, 2 ** 31 - 1, (K // per, N), generator=g, dtype=torch.int32)
        qw = qw ^ (torch.randint(0, 2, (K // per, N), generator=g, dtype=torch.int32) << 31)
        qz = torch.randint(0, 2 ** 31 - 1, (K // G, N // per), generator=g, dtype=torch.int32)
        if bits_ == 4:
            # keep every zero nibble <= 14 so that zero+1 stays a 4-bit value (as real checkpoints do)
            qz = qz & 0x66666666
        sc = (torch.rand(K // G, N, generator=g) * 0.01 + 0.001).to(torch.float16)
        info = ("layer.weight", {"layer.weight": qw, "layer.qzeros": qz, "layer.scales": sc})
        w, s, z = ref.repack_gptq_to_a16wX(info, bits_)
        gcases[f"b{bits_}_qweight"] = qw.numpy()
        gcases[f"b{bits_}_qzeros"] = qz.numpy()
        gcases[f"b{bits_}_scales"] = bits(sc)
        gcases[f"b{bits_}_w"] = w.numpy()
        gcases[f"b{bits_}_s"] = bits(s)
        gcases[f"b{bits_}_z"] = bits(z)
    np.savez_compressed(os.path.join(OUT, "quantizer_gptq.npz"), **gcases)
    print("wrote", len(cases), "IQ arrays and", len(gcases), "GPTQ arrays to", OUT)


if __name__ == "__main__":
    sys.exit(main())
