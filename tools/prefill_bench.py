#!/usr/bin/env python3
"""Prefill attention TFLOP/s (Qwen2-7B heads, causal) vs the gfx950 dense bf16 MFMA peak."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops
n, g, H = 28, 4, 128
Ls = [int(x) for x in sys.argv[1:]] or [512, 2048, 4096, 8192]
for L in Ls:
    qkv = torch.randn(L, (n + 2 * g) * H, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, : n * H], qkv[:, n * H:(n + g) * H], qkv[:, (n + g) * H:]
    out = torch.empty(L, n * H, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.prefill_attn(q, k, v, n, g, H, 0.088, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.prefill_attn(q, k, v, n, g, H, 0.088, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = 4 * n * H * L * L / 2
    print(f"L={L}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s  ({flops/ms/1e9/2500*100:.2f}% of 2.5 PFLOP/s dense bf16)")
