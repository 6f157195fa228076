import sys, os, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import decoder
from tests.tp_loopback_lib import LoopbackComm
batch, nranks, STEPS = 2, 8, 5
cfg = decoder.ModelConfig("tp8-test", hidden=1024, layers=2, n_heads=28, n_kv=4, head_dim=128, inter=1024, vocab=4096)
spec = decoder.QuantSpec(4, 128)
ids0 = np.random.default_rng(nranks * 31 + batch).integers(0, cfg.vocab, batch)
def single():
    model = decoder.build_random_model(cfg, spec, seed=99)
    s0 = decoder.DecodeSession(model, batch, max_len=32, span_len=16, kv_mode="none")
    s0.set_state(ids0, [0] * batch)
    out = []
    for _ in range(STEPS):
        s0.step(); torch.cuda.synchronize(); out.append(s0.logits.cpu().numpy().copy())
    return out
def tp():
    shared = LoopbackComm.Shared(nranks)
    lg = [[] for _ in range(nranks)]
    def worker(rank):
        torch.cuda.set_device(0)
        m = decoder.build_random_model(cfg, spec, seed=99, rank=rank, nranks=nranks)
        s = decoder.DecodeSession(m, batch, max_len=32, span_len=16, kv_mode="none", comm=LoopbackComm(shared, rank, nranks))
        s.set_state(ids0, [0] * batch)
        for _ in range(STEPS):
            s.step(); torch.cuda.synchronize(); lg[rank].append(s.logits.cpu().numpy().copy()); shared.bar.wait()
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]; [t.join() for t in ts]
    return [np.concatenate([lg[r][t] for r in range(nranks)], 1) for t in range(STEPS)]
order = sys.argv[1] if len(sys.argv) > 1 else "st"
runs = {}
if order == "st":
    runs["single_cold"] = single(); runs["tp_first"] = tp(); runs["single_warm"] = single(); runs["tp_second"] = tp()
else:
    runs["tp_first"] = tp(); runs["single_cold"] = single(); runs["tp_second"] = tp(); runs["single_warm"] = single()
for a, b in (("single_cold", "single_warm"), ("tp_first", "tp_second"), ("single_warm", "tp_second"), ("single_cold", "tp_first")):
    print(a, "vs", b, [f"{np.abs(x - y).max():.2e}" for x, y in zip(runs[a], runs[b])])

a, b = runs["tp_first"][0], runs["tp_second"][0]
V = a.shape[1] // nranks
print("step-0 diff per rank slice:", [f"{np.abs(a[:, r*V:(r+1)*V] - b[:, r*V:(r+1)*V]).max():.1e}" for r in range(nranks)])
print("step-0 diff per row:", [f"{np.abs(a[m] - b[m]).max():.1e}" for m in range(batch)])
