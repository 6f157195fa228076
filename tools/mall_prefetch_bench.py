#!/usr/bin/env python3
"""Does a weight-streaming decode GEMV start faster when its weights were pulled into the Infinity Cache by the launch before it?
   rocprofv3 --kernel-trace --stats -- python tools/mall_prefetch_bench.py        (MODE=cold | full | head; one mode per process)
Eight layers' gate/up (72 MB each) and down (36 MB) matrices in rotation, so that every launch finds its own weights cold (576 + 288 MB
against the 256 MB Infinity Cache); MODE=full sweeps the whole matrix with dihip_prefetch in the launch before, MODE=head the first HEAD_MB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _load_pkg

_load_pkg()
from dash_infer_amd import decoder, ops

MODE = os.environ.get("MODE", "cold")
HEAD = int(float(os.environ.get("HEAD_MB", "16")) * 2**20)
cfg = decoder.ModelConfig("mall", hidden=3584, layers=8, n_heads=28, n_kv=4, head_dim=128, inter=18944, vocab=2048)
model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128, gptq_like_zeros=True), seed=5)
s = decoder.DecodeSession(model, 1, max_len=256, span_len=128)
s.h.normal_()


def wt(p):
    return [t for t in (getattr(p, "w", None), getattr(p, "sz", None)) if t is not None]


def head_of(ts):
    out = []
    for t in ts:
        flat = t.view(torch.uint8).view(-1)
        out.append(flat[: min(HEAD, flat.numel())])
    return out


def sweep():
    for lw in model.layers:
        if MODE == "full":
            ops.prefetch(wt(lw.gate) + wt(lw.up), workgroups=256)
        elif MODE == "head":
            ops.prefetch(head_of(wt(lw.gate)) + head_of(wt(lw.up)), workgroups=256)
        ops.fused_norm_swiglu(s.h, lw.ln2, cfg.eps, lw.gate, lw.up, s.scratch, out=s.act)
        if MODE == "full":
            ops.prefetch(wt(lw.down), workgroups=256)
        elif MODE == "head":
            ops.prefetch(head_of(wt(lw.down)), workgroups=256)
        ops.fused_gemm_addto(s.act, lw.down, s.h, s.scratch, out=s.h, M=1)


st = torch.cuda.Stream()
with torch.cuda.stream(st):
    sweep()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    sweep()
for _ in range(40):
    g.replay()
torch.cuda.synchronize()
print("mode", MODE, "done")
