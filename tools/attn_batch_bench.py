#!/usr/bin/env python3
"""Op-boundary decode attention (dihip_span_attn_decode) at batch: time per layer and KV GB/s.
Shapes: BASELINE configs[2] (7B, batch 32, 2048) and configs[3] per rank (72B TP=8: 8 heads / 1 KV head, batch 16, 4096)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops

def run(B, n, g, L, mode, S=128, H=128, layers=8):
    dt = torch.bfloat16
    spr = (L + S - 1) // S + 1
    pool = ops.SpanPool(2 * layers * B * spr + 1, g, S, H, mode, dt, "cuda")
    kvs = [ops.KVCacheSet(pool, B, spr) for _ in range(layers)]
    for kv in kvs:
        for b in range(B):
            kv.ensure(b, L)
        kv.sync()
    # finite contents: bf16 bytes 0x3c.. (around 1.0) / arbitrary quantised bytes with sane (zero, scale) are not needed for timing
    pool.pool.fill_(0x3c)
    q = torch.randn(B, n * H, device="cuda").to(dt)
    lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
    ws = torch.empty(max(ops.span_attn_workspace(B, n, H, L), 256), dtype=torch.uint8, device="cuda")
    sync = torch.zeros(int(ops.lib().dihip_span_attn_sync_bytes(B, n)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, n * H, dtype=dt, device="cuda")
    def sweep():
        for kv in kvs:
            ops.span_attn_decode(q, kv, lens, n, g, H, L, 0.088, ws, sync, out=out)
    for _ in range(2):
        sweep()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        sweep()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * layers)
    kvb = {"none": H * 2, "i8": H + 8, "u4": H // 2 + 8}[mode]
    byts = B * 2 * g * L * kvb
    print(f"B={B} n={n} g={g} L={L} kv={mode}: {us:.1f} us/layer, {byts/1e6:.1f} MB -> {byts/us/1e3:.0f} GB/s")

if len(sys.argv) > 1:   # e.g. "1,4,8,16": batch sweep of the 7B shape with the 16-bit cache
    for B in [int(x) for x in sys.argv[1].split(",")]:
        run(B, 28, 4, 2048, "none")
else:
    for mode in ("none", "i8", "u4"):
        run(32, 28, 4, 2048, mode)
        run(16, 8, 1, 4096, mode)
