#!/usr/bin/env python3
"""Per-wave wall-clock stamps of the uint4-cache decode attention (span_attn_u4_mfma_kernel) at BASELINE configs[2]'s shape:
batch 32, 28 query / 4 KV heads, 2048 cached tokens.  Needs the trace build:  make -C dash-infer_amd/csrc trace  and
DIHIP_LIB_DIR=dash-infer_amd/lib/trace.  Stamps (10 ns ticks): entry, first loads issued + q built, first 32 tokens done,
token loop done, [4] partial record written, [5] ticket taken, [6] merge done (last arriver), end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops
B, n, g, H, L, S = int(os.environ.get("B", 32)), 28, 4, 128, int(os.environ.get("L", 2048)), 128
mode = os.environ.get("KV", "u4")
dt = torch.bfloat16
max_spans = (L + 1 + S - 1) // S + 1
pool = ops.SpanPool(2 * B * max_spans + 4, g, S, H, mode, dt)
pool.pool.random_(0, 256)
kv = ops.KVCacheSet(pool, B, max_spans)
for b in range(B):
    kv.ensure(b, L + 1)
kv.sync()
if mode != "none":   # sane (zero, scale) pairs behind the data of every span
    nb = ops.span_bytes(g, S, H, mode, dt)
    par = pool.pool.view(-1, pool.aligned)[:, g * S * (H // 2 if mode == "u4" else H): nb].view(torch.float32)
    par[:, 0::2] = 8.0
    par[:, 1::2] = 0.01
q = (torch.randn(B, n * H, device="cuda") * 0.5).to(dt)
lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
ws = torch.empty(ops.span_attn_workspace(B, n, H, L + 16), dtype=torch.uint8, device="cuda")
sync = torch.zeros(int(ops.lib().dihip_span_attn_sync_bytes(B, n)), dtype=torch.uint8, device="cuda")
scale = H ** -0.5
run = lambda: ops.span_attn_decode(q, kv, lens, n, g, H, L + 16, scale, ws, sync)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); torch.cuda.synchronize()
print(f"B={B} L={L} kv={mode}: {e0.elapsed_time(e1) * 50:.2f} us per launch")
trace = torch.zeros(64 * g * B * 32, dtype=torch.int64, device="cuda")
ops.lib().dihip_debug_set_trace(trace.data_ptr(), trace.numel() * 8)
run()
torch.cuda.synchronize()
ops.lib().dihip_debug_set_trace(None, 0)
t = trace.cpu().numpy().reshape(-1, 4, 8)
t = t[(t[:, :, 0] != 0).all(axis=1)]
t0 = t[:, :, 0].min()
print(f"{t.shape[0]} workgroups; us after the first wave's entry: min / median / max over waves")
names = ["entry", "loads issued, q built", "first 32 tokens done", "token loop done", "partial written", "ticket taken", "merge done (last arriver)", "end"]
for i, nme in enumerate(names):
    v = t[:, :, i][t[:, :, i] != 0]
    if v.size == 0:
        continue
    v = (v - t0) * 0.01
    print(f"  {nme:28s} {v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}   n={v.size}")
d = (t[:, :, 3] - t[:, :, 0]) * 0.01
print(f"  per wave entry -> loop done: median {np.median(d):.2f} us; entry spread over workgroups {(t[:, 0, 0].max() - t0) * 0.01:.2f} us")
