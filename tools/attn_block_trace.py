#!/usr/bin/env python3
"""Timeline of dihip_decode_attn_block (csrc/decode_attn_block.hip) on the `make trace` build:
   DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace python tools/attn_block_trace.py
Per role (attention workgroups / GEMV workgroups) the wall-clock stamps of wave 0 relative to the launch's first stamp: median and
max over the workgroups, in us (100 MHz counter: 10 ns steps).  Also graph-times the block against the three launches it replaces."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _load_pkg

_load_pkg()
from dash_infer_amd import decoder, ops

L = int(os.environ.get("HIST", "2048"))
cfg = decoder.ModelConfig("trace", hidden=3584, layers=int(os.environ.get("LAYERS", "8")), n_heads=28, n_kv=4, head_dim=128, inter=1024, vocab=2048)
model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128, gptq_like_zeros=True), seed=5)


def session(block):
    os.environ["DIHIP_DECODER_ATTN_BLOCK"] = "1" if block else "0"
    s = decoder.DecodeSession(model, 1, max_len=L + 64, span_len=128)
    s.fill_cache_random(L, seed=3)
    s.set_state([1], [L])
    s.h.normal_()
    return s


def graph_time(s, name):
    def sweep():
        for li in range(len(model.layers)):
            lw = model.layers[li]
            if s.attn_block:
                ops.decode_attn_block(s.h, s.h, lw.ln1, cfg.eps, lw.qkv, lw.qkv_bias, lw.o, s.kv[li], s.old_lens, s.rope_tab, s.n_loc, s.g_loc,
                                      s.H, s.max_len, s.scale, s.attn_ws, s.block_sync, out=s.h)
            else:
                ops.fused_norm_gemm(s.h, lw.ln1, cfg.eps, lw.qkv, lw.qkv_bias, s.scratch, out=s.qkv)
                ops.span_attn_decode_fused(s.qkv, s.kv[li], s.old_lens, s.rope_tab, s.n_loc, s.g_loc, s.H, s.max_len, s.scale, s.attn_ws,
                                           out=s.attn, sync=s.attn_sync)
                ops.fused_gemm_addto(s.attn, lw.o, s.h, s.scratch, out=s.h, M=1)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sweep()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sweep()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / (20 * len(model.layers)):6.2f} us per layer (graph replay, {len(model.layers)} layers' weights)")


a, b = session(False), session(True)
print("attn_block served:", b.attn_block)
graph_time(a, "three launches")
graph_time(b, "one launch    ")
if "trace" in os.environ.get("DIHIP_LIB_DIR", ""):
    nwg = 256
    tr = torch.zeros(nwg * 32, dtype=torch.int64, device="cuda")
    ops.lib().dihip_debug_set_trace(ops.ptr(tr), tr.numel() * 8)
    lw = model.layers[0]
    for rep in range(3):
        tr.zero_()
        ops.decode_attn_block(b.h, b.h, lw.ln1, cfg.eps, lw.qkv, lw.qkv_bias, lw.o, b.kv[0], b.old_lens, b.rope_tab, b.n_loc, b.g_loc, b.H, b.max_len,
                              b.scale, b.attn_ws, b.block_sync, out=b.h)
        torch.cuda.synchronize()
    ops.lib().dihip_debug_set_trace(None, 0)
    t = tr.view(nwg, 32).cpu().double()
    used = t[:, 0] > 0
    t0 = t[used][:, :].clone()
    base = t0[t0 > 0].min()
    import ctypes
    ns_c, mf_c = ctypes.c_int(0), ctypes.c_int(0)
    ops.lib().dihip_debug_attn_plan(1, cfg.n_heads, cfg.n_kv, b.max_len, 0, 2, 0, ctypes.byref(ns_c), ctypes.byref(mf_c))  # 16-bit cache, bf16
    ns, g = ns_c.value, cfg.n_kv
    na = int(os.environ.get("NA", ns * g))
    print(f"plan: {ns} splits x {g} groups = {na} attention workgroups")
    names_a = ["entry", "K/V requested", "q gathered (+rotate)", "tiles done", "records drained", "ticket taken", "merge loads landed", "end"]
    names_a[4] = "records stored (dist. merge) / drained"
    names_g = ["entry", "row staged, shares requested", "shares landed", "qkv published", "group sentinels seen", "output swept", "o multiplied", "end",
               "record slices seen (dist. merge)", "merged elements published"]
    def show(rows, names, title):
        print(title, f"({rows.shape[0]} workgroups; us after the launch's first stamp: median / max)")
        for i, nm in enumerate(names):
            col = rows[:, i]
            col = col[col > 0]
            if col.numel():
                rel = (col - base) / 100.0
                print(f"  {i} {nm:32s} {rel.median().item():7.2f} {rel.max().item():7.2f}   (n={col.numel()})")
    show(t[:na][used[:na]][:, :8], names_a, "attention workgroups, wave 0")
    if os.environ.get("BY_SPLIT"):
        # per split index (max over the groups): which workgroup is the late one?
        ta = t[:na].view(g, ns, 32)
        for col, nm in ((2, "q gathered"), (3, "tiles done"), (4, "records stored"), (6, "merge loads landed"), (7, "end")):
            v = (ta[:, :, col].max(dim=0).values - base) / 100.0
            print(f"  by split, {nm:20s}: " + " ".join(f"{x:5.2f}" for x in v.tolist()))
    show(t[:na][used[:na]][:, 24:32], ["V tile in LDS", "scores done", "softmax done", "loop left", "granules swept"], "attention workgroups, wave 0, inside the tile")
    show(t[na:][used[na:]][:, :10], names_g, "GEMV workgroups, wave 0 (stamps 8, 9 come between 3 and 4)")
