#!/usr/bin/env python3
"""Decode-step attention of the 16-bit cache at batch 1 / 4 (headline shape: 28 heads, 4 KV heads, 2048 tokens), graph-timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops

def run(B, n, g, L, mode="none", S=128, H=128, layers=8):
    dt = torch.bfloat16
    spr = (L + 2 + S - 1) // S + 1
    pool = ops.SpanPool(2 * layers * B * spr + 1, g, S, H, mode, dt, "cuda")
    kvs = [ops.KVCacheSet(pool, B, spr) for _ in range(layers)]
    for kv in kvs:
        for b in range(B):
            kv.ensure(b, L + 2)
        kv.sync()
    pool.pool.fill_(0x3c)
    qkv = torch.randn(B, (n + 2 * g) * H, device="cuda").to(dt)
    old = torch.full((B,), L, dtype=torch.int32, device="cuda")
    inv = torch.tensor([1.0 / (1e6 ** (2 * i / H)) for i in range(H // 2)], dtype=torch.float32, device="cuda")
    tab = ops.rope_table(inv, L + 8, H)
    ws = torch.empty(max(ops.span_attn_workspace(B, n, H, L + 2), ops.span_attn_fused_workspace(B, n, g, H, L + 2), 256), dtype=torch.uint8, device="cuda")
    sync = torch.zeros(int(ops.lib().dihip_span_attn_sync_bytes(B, n)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, n * H, dtype=dt, device="cuda")
    def sweep():
        for kv in kvs:
            ops.span_attn_decode_fused(qkv, kv, old, tab, n, g, H, L + 2, 0.088, ws, out=out, sync=sync)
    for _ in range(2):
        sweep()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        sweep()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} n={n} g={g} L={L} kv={mode}  decode-step attention {e0.elapsed_time(e1) * 1e3 / (10 * layers):6.2f} us/layer")

for B in (1, 4):
    run(B, 28, 4, 2048)
