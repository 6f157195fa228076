#!/usr/bin/env python3
"""Stand-alone timing of the deferred-RMSNorm consumer / producer kernels (round 5): the SwiGLU pair with and without row partials on
the same input, and the residual GEMM with the norm deferred / as a separate launch.  Qwen2-7B widths, int4 g128.
    python tools/defer_norm_bench.py [M]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_pkg
load_pkg()
from dash_infer_amd import decoder, ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = decoder.ModelConfig("b", hidden=3584, layers=4, n_heads=28, n_kv=4, head_dim=128, inter=18944, vocab=1024)
model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=1)
L = model.layers
sc = ops.Scratch(ops.lowp_workspace_bytes(4, M, 18944, 3584, 128) * 2)
dev = "cuda"
clay = ops.ACT_FRAG32 if ops.prefers_frag(L[0].gate, M, dual=True) else ops.ACT_ROWMAJOR
xlay = ops.ACT_FRAG32 if ops.prefers_frag(L[0].o, M) else ops.ACT_ROWMAJOR
attn = torch.randn(M, 3584, device=dev).bfloat16()
attn_in = ops.act_to_frag(attn) if xlay == ops.ACT_FRAG32 else attn
h = torch.randn(M, 3584, device=dev)
hout = torch.empty_like(h)
xn = torch.zeros(ops.act_frag_numel(M, 3584) if clay == ops.ACT_FRAG32 else M * 3584, dtype=torch.bfloat16, device=dev)
rowsq = ops.rowsq_buffer(dev)
act = torch.zeros(ops.act_frag_numel(M, 18944), dtype=torch.bfloat16, device=dev)


def timed(fn, n=300):
    for _ in range(10):
        fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                fn(i)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000 / n)
    return min(ts)


_, parts = ops.fused_gemm_addto_prenorm(attn_in, L[0].o, h, sc, L[0].ln2, 1e-6, xn, rowsq, out=hout, x_layout=xlay, xnorm_layout=clay, M=M)
print(f"M = {M}: parts {parts}, consumer layout {'FRAG32' if clay == ops.ACT_FRAG32 else 'row-major'}")
t0 = timed(lambda i: ops.prenorm_swiglu(xn, L[i % 4].gate, L[i % 4].up, sc, M, x_layout=clay, out=act, y_layout=ops.ACT_FRAG32))
t1 = float("nan") if os.environ.get("PLAIN_ONLY") else timed(lambda i: ops.prenorm_swiglu_rowsq(xn, L[i % 4].gate, L[i % 4].up, sc, M, rowsq, parts, 1e-6, x_layout=clay, out=act, y_layout=ops.ACT_FRAG32))
print(f"SwiGLU pair alone: plain {t0:.2f} us, with row partials {t1:.2f} us")
if os.environ.get("PLAIN_ONLY"):
    sys.exit(0)
p0 = timed(lambda i: ops.fused_gemm_addto_norm(attn_in, L[i % 4].o, h, sc, L[i % 4].ln2, 1e-6, xn, out=hout, x_layout=xlay, xnorm_layout=clay, M=M))
p1 = timed(lambda i: ops.fused_gemm_addto_prenorm(attn_in, L[i % 4].o, h, sc, L[i % 4].ln2, 1e-6, xn, rowsq, out=hout, x_layout=xlay, xnorm_layout=clay, M=M))
print(f"o-projection + residual: with the norm launch {p0:.2f} us, norm deferred {p1:.2f} us")


def chain(defer):
    def f(i):
        lw = L[i % 4]
        if defer:
            _, pp = ops.fused_gemm_addto_prenorm(attn_in, lw.o, h, sc, lw.ln2, 1e-6, xn, rowsq, out=hout, x_layout=xlay, xnorm_layout=clay, M=M)
            ops.prenorm_swiglu_rowsq(xn, lw.gate, lw.up, sc, M, rowsq, pp, 1e-6, x_layout=clay, out=act, y_layout=ops.ACT_FRAG32)
        else:
            ops.fused_gemm_addto_norm(attn_in, lw.o, h, sc, lw.ln2, 1e-6, xn, out=hout, x_layout=xlay, xnorm_layout=clay, M=M)
            ops.prenorm_swiglu(xn, lw.gate, lw.up, sc, M, x_layout=clay, out=act, y_layout=ops.ACT_FRAG32)
    return f


c0, c1 = timed(chain(False)), timed(chain(True))
print(f"o-projection -> SwiGLU pair: three launches {c0:.2f} us, two launches (norm deferred) {c1:.2f} us")
