"""Runs the context-phase SwiGLU GEMM (Qwen2-7B gate/up, int4 g128, M = 2048) a few times: the target of rocprofv3 PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import torch
pkg = importlib.import_module("dash-infer_amd")
decoder = importlib.import_module("dash-infer_amd.decoder")
ops = importlib.import_module("dash-infer_amd.ops")
cfg = decoder.QWEN2_7B
L = int(os.environ.get("M", "2048"))
model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128, gptq_like_zeros=True), seed=1234, layers=1)
lw = model.layers[0]
sc = ops.Scratch(max(ops.lowp_workspace_bytes(4, L, p.N, p.K, 128) for p in (lw.qkv, lw.o, lw.gate, lw.down)), "cuda")
h = torch.randn(L, cfg.hidden, device="cuda") * 0.5
for _ in range(int(os.environ.get("REPS", "6"))):
    ops.fused_norm_swiglu(h, lw.ln2, cfg.eps, lw.gate, lw.up, sc)
torch.cuda.synchronize()
print("done")
