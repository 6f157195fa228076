import sys, os, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import decoder
from tests.tp_loopback_lib import LoopbackComm
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nranks = 8
cfg = decoder.ModelConfig("tp8-test", hidden=1024, layers=2, n_heads=28, n_kv=4, head_dim=128, inter=1024, vocab=4096)
spec = decoder.QuantSpec(4, 128)
ids0 = np.random.default_rng(nranks * 31 + batch).integers(0, cfg.vocab, batch)
STEPS = 5

def instrument(sess, log):
    orig = sess._proj_residual
    def wrapped(*a, **k):
        r = orig(*a, **k)
        torch.cuda.synchronize()
        log.append(sess.h.cpu().numpy().copy())
        return r
    sess._proj_residual = wrapped

model = decoder.build_random_model(cfg, spec, seed=99)
s0 = decoder.DecodeSession(model, batch, max_len=32, span_len=16, kv_mode="none")
s0.set_state(ids0, [0] * batch)
log0 = []
instrument(s0, log0)
L0 = []
for _ in range(STEPS):
    s0.step(); torch.cuda.synchronize(); L0.append((s0.logits.cpu().numpy().copy(), s0.ids.cpu().numpy().copy()))
shared = LoopbackComm.Shared(nranks)
logs = [[] for _ in range(nranks)]
lg = [[] for _ in range(nranks)]
def worker(rank):
    torch.cuda.set_device(0)
    m = decoder.build_random_model(cfg, spec, seed=99, rank=rank, nranks=nranks)
    s = decoder.DecodeSession(m, batch, max_len=32, span_len=16, kv_mode="none", comm=LoopbackComm(shared, rank, nranks))
    s.set_state(ids0, [0] * batch)
    instrument(s, logs[rank])
    for _ in range(STEPS):
        s.step(); torch.cuda.synchronize(); lg[rank].append((s.logits.cpu().numpy().copy(), s.ids.cpu().numpy().copy())); shared.bar.wait()
ts = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
[t.start() for t in ts]; [t.join() for t in ts]
for i, (a, b) in enumerate(zip(log0, logs[0])):
    print(f"after residual GEMM {i} (step {i // 4}, layer {(i // 2) % 2}, {'o' if i % 2 == 0 else 'down'}): max |single - tp| = {np.abs(a - b).max():.3e}  (|h| max {np.abs(a).max():.2f})") if np.abs(a - b).max() > 1e-4 else None
for t in range(STEPS):
    lo = np.concatenate([lg[r][t][0] for r in range(nranks)], 1)
    print("step", t, "logits diff", np.abs(lo - L0[t][0]).max(), "ids single", L0[t][1], "tp", lg[0][t][1])
