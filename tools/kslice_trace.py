#!/usr/bin/env python3
"""Per-wave timeline of the K-slice small-batch GEMM (gemm_kslice_kernel.hpp) on the Qwen2-7B gate/up pair:
wall-clock stamps (10 ns ticks) written by lane 0 of every wave -- entry, activation loads issued, activations landed +
ring issued, first tile pair streamed, first barrier passed, end.
Needs the library built with the stamps: make -C dash-infer_amd/csrc CXXFLAGS_EXTRA=-DDIHIP_KSL_TRACE (off by default)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K, N, G, wbits = 3584, 18944, 128, 4
gen = torch.Generator(device="cuda").manual_seed(1)
def rand_packed():
    q = torch.randint(0, 256, (K, N // 2), dtype=torch.uint8, device="cuda", generator=gen)
    s = torch.full((K // G, N), 0.01, dtype=torch.bfloat16, device="cuda")
    z = torch.full((K // G, N), 8.0, dtype=torch.bfloat16, device="cuda")
    return ops.pack_lowp(q, s, z, G, wbits)
pg, pu = rand_packed(), rand_packed()
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
xf = ops.act_to_frag(x)
sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
for _ in range(3):
    y = ops.prenorm_swiglu(xf, pg, pu, sc, M, x_layout=ops.ACT_FRAG32)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.prenorm_swiglu(xf, pg, pu, sc, M, x_layout=ops.ACT_FRAG32)
e1.record(); torch.cuda.synchronize()
print(f"M={M}: {e0.elapsed_time(e1) * 100:.1f} us per launch (back to back, L2/MALL-warm weights)")
trace = torch.zeros(512 * 8 * 8, dtype=torch.int64, device="cuda")
ops.lib().dihip_debug_set_trace(trace.data_ptr(), trace.numel() * 8)
ops.prenorm_swiglu(xf, pg, pu, sc, M, x_layout=ops.ACT_FRAG32)
torch.cuda.synchronize()
ops.lib().dihip_debug_set_trace(None, 0)
t = trace.cpu().numpy().reshape(-1, 8, 8)
t = t[(t[:, :, 0] != 0).any(axis=1)]
t0 = t[:, :, 0][t[:, :, 0] != 0].min()
names = ["entry", "x loads issued", "x landed, ring issued", "first pair streamed", "first barrier passed", "end"]
print(f"{t.shape[0]} workgroups; stamps in us after the first wave's entry: min / median / max over waves")
for i, nme in enumerate(names):
    v = t[:, :, i][t[:, :, i] != 0]
    v = (v - t0) * 0.01
    print(f"  {nme:24s} {v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}")
