#!/usr/bin/env python3
"""uint4 decode-step attention at batch 32 (BASELINE configs[2]): the attention kernel alone, the append launch + attention, and the
one-launch step form (dihip_span_attn_decode_step), graph-timed over 8 layers' caches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops

def run(B, n, g, L, mode="u4", S=128, H=128, layers=8):
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
    q = torch.empty(B, n * H, dtype=dt, device="cuda")
    old = torch.full((B,), L, dtype=torch.int32, device="cuda")
    new = old + 1
    inv = torch.tensor([1.0 / (1e6 ** (2 * i / H)) for i in range(H // 2)], dtype=torch.float32, device="cuda")
    tab = ops.rope_table(inv, L + 8, H)
    ws = torch.empty(max(ops.span_attn_workspace(B, n, H, L + 2), ops.span_attn_fused_workspace(B, n, g, H, L + 2), 256), dtype=torch.uint8, device="cuda")
    sync = torch.zeros(int(ops.lib().dihip_span_attn_sync_bytes(B, n)), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, n * H, dtype=dt, device="cuda")
    forms = {
        "attention alone": lambda kv: ops.span_attn_decode(q, kv, new, n, g, H, L + 2, 0.088, ws, sync, out=out),
        "append + attention": lambda kv: (ops.rope_kv_append(kv, q, qkv, old, inv, n, g, H), ops.span_attn_decode(q, kv, new, n, g, H, L + 2, 0.088, ws, sync, out=out)),
        "append alone": lambda kv: ops.rope_kv_append(kv, q, qkv, old, inv, n, g, H),
        "one-launch step": lambda kv: ops.span_attn_decode_step(qkv, kv, old, tab, n, g, H, L + 2, 0.088, ws, sync, out=out),
    }
    for name, fn in forms.items():
        def sweep():
            for kv in kvs:
                fn(kv)
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
        print(f"B={B} n={n} g={g} L={L} kv={mode}  {name:20s} {e0.elapsed_time(e1) * 1e3 / (5 * layers):6.2f} us/layer")

run(32, 28, 4, 2048)
run(1, 28, 4, 2048)
