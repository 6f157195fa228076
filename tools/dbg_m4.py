import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import ops
from oracle import gemm_ref, quant
from oracle.numerics import bf16_round
K, N, G, wbits = 3584, 18944, 128, 4
rng = np.random.default_rng(9)
x = bf16_round(rng.uniform(-1, 1, (4, K)).astype(np.float32))
W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
q, s, z = quant.iq_quantize_a16w4(W, G, "bf16")
dev = lambda a, dt=torch.bfloat16: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dt).cuda()
pw = ops.pack_lowp(torch.from_numpy(q).cuda(), dev(s), dev(z), G, wbits)
xd = dev(x)
print("plan M=4", ops.gemv_plan(wbits, 4, N, K, G), "M=1", ops.gemv_plan(wbits, 1, N, K, G))
ref = gemm_ref.gemm_a16wx(x, q, s, z, G, wbits, ft="bf16")
for M in (4, 1, 2):
    ys = [ops.gemm_lowp(xd[:M].contiguous(), pw).float().cpu().numpy() for _ in range(4)]
    for i, y in enumerate(ys):
        bad = np.abs(y - ref[:M]) > 0.02 * np.abs(ref[:M]).max()
        cols = np.unique(np.nonzero(bad)[1])
        tiles = np.unique(cols // 16)
        print(f"M={M} run {i}: {bad.sum()} bad values, tiles {tiles[:12]}{'...' if len(tiles) > 12 else ''} (n={len(tiles)}), blocks {np.unique(tiles % 237)[:12]}, unit idx {np.unique(tiles // 237)}")
