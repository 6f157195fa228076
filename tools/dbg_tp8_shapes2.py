import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops
from oracle import gemm_ref, quant, glue
from oracle.numerics import bf16_round
dev = lambda a, dt=torch.bfloat16: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dt).cuda()
rng = np.random.default_rng(0)
G, wbits = 128, 4
for (K, N) in [(512, 1024), (384, 1024), (128, 1024)]:
    W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
    q, s, z = quant.iq_quantize_a16w4(W, G, "bf16")
    pw = ops.pack_lowp(torch.from_numpy(q).cuda(), dev(s), dev(z), G, wbits)
    for M in (1, 2, 3, 4, 5):
        x = bf16_round(rng.normal(0, 1, (M, K)).astype(np.float32))
        sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
        out = torch.full((M, N), 7.0, dtype=torch.float32, device="cuda")
        y2 = ops.fused_gemm_addto(dev(x), pw, None, sc, out=out, M=M).cpu().numpy()
        ref2 = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, ft="f32", round_out=False)
        print(f"addto(no residual) K={K} N={N} M={M}: {np.abs(y2 - ref2).max() / np.abs(ref2).max():.2e}", flush=True)
K, V = 1024, 512
Wl = bf16_round(rng.normal(0, 0.05, (K, V)).astype(np.float32))
lm = ops.pack_dense(dev(Wl))
for M in (1, 2, 3, 4, 5):
    h = rng.normal(0, 1, (M, K)).astype(np.float32)
    gam = bf16_round(1 + rng.normal(0, 0.1, K).astype(np.float32))
    sc = ops.Scratch(max(int(ops.lib().dihip_dense_workspace_bytes(M, V, K)), 1024))
    lo = ops.lm_head(dev(h, torch.float32), dev(gam), 1e-6, lm, sc).cpu().numpy()
    xn = bf16_round(glue.rmsnorm(h, gam, 1e-6))
    ref = (xn.astype(np.float64) @ Wl.astype(np.float64)).astype(np.float32)
    print(f"lm_head V={V} M={M}: {np.abs(lo - ref).max() / np.abs(ref).max():.2e}", flush=True)
