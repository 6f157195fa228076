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
for (K, N) in [(1024, 768), (1024, 640), (512, 1024), (384, 1024), (128, 1024), (1024, 128)]:
    W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
    q, s, z = quant.iq_quantize_a16w4(W, G, "bf16")
    pw = ops.pack_lowp(torch.from_numpy(q).cuda(), dev(s), dev(z), G, wbits)
    W2 = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
    q2, s2, z2 = quant.iq_quantize_a16w4(W2, G, "bf16")
    pw2 = ops.pack_lowp(torch.from_numpy(q2).cuda(), dev(s2), dev(z2), G, wbits)
    for M in (1, 2, 3, 4, 5):
        x = bf16_round(rng.normal(0, 1, (M, K)).astype(np.float32))
        sc = ops.Scratch(ops.lowp_workspace_bytes(wbits, M, N, K, G))
        y = ops.gemm_lowp(dev(x), pw, scratch=sc).float().cpu().numpy()
        ref = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, ft="bf16")
        e1 = np.abs(y - ref).max() / np.abs(ref).max()
        h = rng.normal(0, 1, (M, K)).astype(np.float32)
        gam = bf16_round(1 + rng.normal(0, 0.1, K).astype(np.float32))
        hres = rng.normal(0, 1, (M, N)).astype(np.float32)
        y2 = ops.fused_gemm_addto(dev(x), pw, dev(hres, torch.float32), sc, M=M).cpu().numpy()
        ref2 = hres + gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, ft="f32", round_out=False)
        e2 = np.abs(y2 - ref2).max() / np.abs(ref2).max()
        y3 = ops.fused_norm_gemm(dev(h, torch.float32), dev(gam), 1e-6, pw, None, sc).float().cpu().numpy()
        xn = bf16_round(glue.rmsnorm(h, gam, 1e-6))
        ref3 = gemm_ref.gemm_a16wx(xn, q, s, z, G, wbits, ft="bf16")
        e3 = np.abs(y3 - ref3).max() / np.abs(ref3).max()
        y4 = ops.fused_norm_swiglu(dev(h, torch.float32), dev(gam), 1e-6, pw, pw2, sc).float().cpu().numpy()
        gg = gemm_ref.gemm_a16wx(xn, q, s, z, G, wbits, ft="f32", round_out=False)
        uu = gemm_ref.gemm_a16wx(xn, q2, s2, z2, G, wbits, ft="f32", round_out=False)
        ref4 = bf16_round(glue.silu(gg) * uu)
        e4 = np.abs(y4 - ref4).max() / max(np.abs(ref4).max(), 1e-9)
        flag = "  <<<<" if max(e1, e2, e3, e4) > 1.2e-2 else ""
        print(f"K={K:5d} N={N:5d} M={M}: plain {e1:.2e}  addto {e2:.2e}  norm {e3:.2e}  swiglu {e4:.2e}  plan {ops.gemv_plan(wbits, M, N, K, G)}{flag}", flush=True)
