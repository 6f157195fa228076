#!/usr/bin/env python3
"""Mixture-of-experts decode step (route + expert GEMVs + combine) at the Qwen2-57B-A14B shape of BASELINE configs[4]:
64 experts, top-8, hidden 3584, expert width 2560, int8 per-channel weight-only.  Reports us per layer and the
weight bytes streamed per second (each (token, expert) slot reads 3 * hidden * proj bytes + scales)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops

E, k, hidden, proj, wbits = 64, 8, 3584, 2560, 8
layers = 4
dt = torch.bfloat16
def experts(N, K):
    q = torch.randint(-128, 128, (K, N), dtype=torch.int8, device="cuda")
    s = (torch.rand(1, N, device="cuda") * 0.002 + 0.007).to(dt)
    z = (torch.rand(1, N, device="cuda") * 4 - 2).to(dt)
    return ops.pack_experts([q] * E, [s] * E, [z] * E, -1, wbits)
stacks = [(experts(proj, hidden), experts(proj, hidden), experts(hidden, proj)) for _ in range(layers)]
for T in (1, 4, 16, 32):
    x = torch.randn(T, hidden, device="cuda").to(dt)
    logits = torch.randn(T, E, device="cuda").to(dt)
    ws = torch.empty(int(ops.lib().dihip_moe_workspace_bytes(T, k, hidden, proj)), dtype=torch.uint8, device="cuda")
    out = torch.empty(T, hidden, dtype=dt, device="cuda")
    def sweep():
        for g, u, d in stacks:
            sc, ex = ops.moe_route(logits, k)
            ops.moe_experts(x, ex, sc, g, u, d, ws=ws, out=out)
    sweep(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        sweep()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * layers)
    byts = T * k * 3 * hidden * proj          # upper bound: every slot a different expert
    print(f"T={T}: {us:.1f} us/layer; {T*k} slots, {byts/1e6:.0f} MB of expert weights -> {byts/us/1e3:.0f} GB/s")
