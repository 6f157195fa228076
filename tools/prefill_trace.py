#!/usr/bin/env python3
"""Per-wave cycle stamps of the context-phase GEMM (gemm_prefill_kernel.hpp) on the Qwen2-7B gate/up pair at M = 2048: k-tiles 8..11
of every workgroup -- loop top, after each of the four k-steps' MFMAs were issued, after the fix-up, after staging, after the barrier.
Needs the library built with the stamps:  make -C dash-infer_amd/csrc VARIANT=pf_trace CXXFLAGS_EXTRA=-DDIHIP_PF_TRACE=1  and
DIHIP_LIB_DIR=dash-infer_amd/lib/pf_trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K, N, G, wbits = 3584, 18944, 128, 4
gen = torch.Generator(device="cuda").manual_seed(1)
def rand_packed():
    q = torch.randint(0, 256, (K, N // 2), dtype=torch.uint8, device="cuda", generator=gen)
    s = torch.full((K // G, N), 0.01, dtype=torch.bfloat16, device="cuda")
    z = torch.full((K // G, N), 8.0, dtype=torch.bfloat16, device="cuda")
    return ops.pack_lowp(q, s, z, G, wbits)
pg, pu = rand_packed(), rand_packed()
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
for _ in range(3):
    ops.prenorm_swiglu(x, pg, pu, sc, M)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.prenorm_swiglu(x, pg, pu, sc, M)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"M={M}: {us:.1f} us per launch = {2.0 * M * 2 * N * K / us / 1e6:.0f} TFLOP/s")
nwg = ((M + 127) // 128) * ((N // 16 + 7) // 8)
trace = torch.zeros(nwg * 8 * 4 * 8, dtype=torch.int64, device="cuda")
ops.lib().dihip_debug_set_trace(trace.data_ptr(), trace.numel() * 8)
ops.prenorm_swiglu(x, pg, pu, sc, M)
torch.cuda.synchronize()
ops.lib().dihip_debug_set_trace(None, 0)
t = trace.cpu().numpy().reshape(nwg, 8, 4, 8)
ok = (t != 0).all(axis=(1, 2, 3))
t = t[ok].astype(np.int64)
print(f"{t.shape[0]} of {nwg} workgroups stamped; shader-clock cycles, median / p10 / p90 over (workgroup, wave, k-tile)")
names = ["loop top -> k-step 0 issued", "k-step 1 issued", "k-step 2 issued", "k-step 3 issued", "fix-up done", "staged", "barrier passed"]
d = np.diff(t, axis=3).reshape(-1, 7)
for i, nme in enumerate(names):
    v = d[:, i]
    print(f"  {nme:30s} {np.median(v):8.0f} {np.percentile(v, 10):8.0f} {np.percentile(v, 90):8.0f}")
per = (t[:, :, 1:, 0] - t[:, :, :-1, 0]).reshape(-1)
print(f"  k-tile period                  {np.median(per):8.0f} {np.percentile(per, 10):8.0f} {np.percentile(per, 90):8.0f}   (matrix pipe: 2 waves x 68 MFMA x 16 = 2176)")
skew = t[:, :, :, 0].max(axis=1) - t[:, :, :, 0].min(axis=1)
print(f"  loop-top skew between the 8 waves of a workgroup: median {np.median(skew):.0f}, p90 {np.percentile(skew, 90):.0f}")
for w in range(8):
    v = (t[:, w, :, 7] - t[:, w, :, 6]).reshape(-1)
    print(f"  wave {w}: barrier wait median {np.median(v):6.0f}   multiply {np.median((t[:, w, :, 4] - t[:, w, :, 0]).reshape(-1)):6.0f}   fix-up+stage {np.median((t[:, w, :, 6] - t[:, w, :, 4]).reshape(-1)):6.0f}")
