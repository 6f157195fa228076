"""debug: per-row error of the small-batch GEMV against the oracle GEMM (run on the GPU box)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops
from oracle import gemm_ref
from test_gpu_gemm import make_case, to_dev
for wbits, G, N, K in [(8, 64, 320, 640), (4, 64, 320, 640), (8, -1, 320, 640), (8, 64, 320, 128), (8, 128, 3584, 3584)]:
    for M in (1, 2, 3, 4):
        rng = np.random.default_rng(M * 7 + N + K + wbits)
        x, q, s, z = make_case(rng, M, N, K, G, wbits, "bf16", style="iq")
        ref = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, alpha=1.0, ft="bf16")
        pw = ops.pack_lowp(to_dev(q), to_dev(s, "bf16"), to_dev(z, "bf16"), G, wbits)
        y = ops.gemm_lowp(to_dev(x, "bf16"), pw, alpha=1.0)
        torch.cuda.synchronize()
        err = np.abs(y.float().cpu().numpy() - ref).max(axis=1)
        print(f"w{wbits} G{G} N{N} K{K} M{M}: row max err", np.round(err, 4), "ref max", np.abs(ref).max().round(3), flush=True)
