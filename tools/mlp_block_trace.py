#!/usr/bin/env python3
"""Timeline of dihip_decode_mlp_block (csrc/decode_mlp_block.hip) on the `make trace` build:
   DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace python tools/mlp_block_trace.py
Per phase (gate/up + SwiGLU producers, down-projection consumers) the per-wave stamps of gemv_stream_body relative to the launch's
first stamp: min / median / max over all waves, in us.  Also graph-times the block against the two launches it replaces."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _load_pkg

_load_pkg()
from dash_infer_amd import decoder, ops

LAYERS = int(os.environ.get("LAYERS", "8"))
cfg = decoder.ModelConfig("trace", hidden=3584, layers=LAYERS, n_heads=28, n_kv=4, head_dim=128, inter=18944, vocab=2048)
model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128, gptq_like_zeros=True), seed=5)
l0 = model.layers[0]
sc = ops.Scratch(max(ops.lowp_workspace_bytes(4, 1, p.N, p.K, 128) for p in (l0.gate, l0.down)))
sync = torch.zeros(int(ops.lib().dihip_decode_mlp_block_sync_bytes(cfg.inter)), dtype=torch.uint8, device="cuda")
h = torch.randn(1, cfg.hidden, device="cuda", dtype=torch.float32)
act = torch.empty(1, cfg.inter, dtype=torch.bfloat16, device="cuda")
out = torch.empty_like(h)


def graph_time(fn, name):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for lw in model.layers:
            fn(lw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for lw in model.layers:
            fn(lw)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / (20 * LAYERS):6.2f} us per layer (graph replay, {LAYERS} layers' weights)")


def chain(lw):
    ops.fused_norm_swiglu(h, lw.ln2, cfg.eps, lw.gate, lw.up, sc, out=act)
    ops.fused_gemm_addto(act, lw.down, h, sc, out=out, M=1)


def block(lw):
    ops.decode_mlp_block(h, h, lw.ln2, cfg.eps, lw.gate, lw.up, lw.down, sync, out=out)


graph_time(chain, "two launches")
graph_time(block, "one launch  ")
if "trace" in os.environ.get("DIHIP_LIB_DIR", ""):
    nb = 256
    per = nb * 8 * 8
    tr = torch.zeros(2 * per, dtype=torch.int64, device="cuda")
    names = ["entry", "ring issued", "early landed / flags awaited", "prologue done (row staged)", "main loop done", "block synced", "end", "early issued"]
    for mode, fn in (("ONE launch", block), ("two launches", chain)):
        ops.lib().dihip_debug_set_trace(ops.ptr(tr), tr.numel() * 8)
        for rep in range(3):
            tr.zero_()
            fn(model.layers[rep % LAYERS])
            torch.cuda.synchronize()
        ops.lib().dihip_debug_set_trace(None, 0)
        t = tr.cpu().double()
        if mode == "two launches":   # each launch stamps [workgroup][wave][8] from the start of the buffer: only the last (down) survives
            phases = [("down launch alone", t[:per].view(nb, 8, 8))]
        else:
            phases = [("phase 1: gate/up + SwiGLU", t[:per].view(nb, 8, 8)), ("phase 2: down", t[per:].view(nb, 8, 8))]
        base = min(p[p > 0].min() for _, p in phases if (p > 0).any())
        print(f"== {mode}: stamps in us after the first wave's entry (min / median / max over waves)")
        for title, p in phases:
            print(" ", title)
            for i in (0, 7, 1, 2, 3, 4, 5, 6):
                col = p[:, :, i].flatten()
                col = col[col > 0]
                if col.numel():
                    rel = (col - base) / 100.0
                    print(f"    {names[i]:34s} {rel.min().item():7.2f} {rel.median().item():7.2f} {rel.max().item():7.2f}   (n={col.numel()})")
