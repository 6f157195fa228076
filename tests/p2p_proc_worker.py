"""One rank PROCESS of tests/test_gpu_p2p_processes.py: the one-shot peer-to-peer all-reduce between processes that share
GPU 0 (IPC handles exchanged over a gloo process group; RCCL refuses two ranks on one device, so the fallback communicator
here is torch.distributed).  usage: p2p_proc_worker.py <rank> <world> <port>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from __graft_entry__ import _load_pkg  # noqa: E402

_load_pkg()
from dash_infer_amd import decoder  # noqa: E402

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    comm = decoder.P2PComm(rank, world, dev, decoder.TorchComm(rank, world), guarded=True)
except decoder.P2PUnavailable as e:
    print(f"P2P_UNAVAILABLE rank {rank}: {e}", flush=True)
    sys.exit(3)
gen = torch.Generator(device="cpu").manual_seed(5)
scale = world * (world + 1) / 2
for dt, n in ((torch.bfloat16, 3584), (torch.float32, 2 * 3584), (torch.float16, 4 * 8192), (torch.float32, 8)):
    base = torch.randint(-8, 9, (n,), generator=gen).float()       # small integers: every partial sum is exact in bf16
    for rep in range(25):
        t = (base * (rank + 1) + rep).to(dt).to(dev)
        comm.allreduce_(t)
        torch.cuda.synchronize()
        want = (base * scale + rep * world).to(dt)
        assert torch.equal(t.cpu(), want), f"rank {rank} {dt} n={n} rep {rep}: max diff {(t.cpu().float() - want.float()).abs().max()}"
# the all-reduce inside a captured hipGraph, replayed (what `bench.py --gpus N` does: the epoch lives on the device, so a
# replay continues the protocol where the last one stopped): 6 all-reduces per graph, 5 replays with fresh inputs
static = torch.zeros(3584, dtype=torch.bfloat16, device=dev)
base = torch.randint(-8, 9, (3584,), generator=gen).float()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):  # warm-up on the capture stream (both ranks: the calls pair up)
        comm.allreduce_(static)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
dist.barrier()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(6):
        comm.allreduce_(static)
for rep in range(5):
    static.copy_((base * (rank + 1) + rep).to(torch.bfloat16))
    torch.cuda.synchronize()
    dist.barrier()
    g.replay()
    torch.cuda.synchronize()
    # six in-place sums: x -> scale * x, six times; exact while it stays a small integer times a power of the scale?  keep it
    # checkable: compare with the same recurrence in f32 rounded to bf16 after every sum, as the kernel does
    want = (base * (rank + 1) + rep).to(torch.bfloat16).float()
    allx = [(base * (r + 1) + rep).to(torch.bfloat16).float() for r in range(world)]
    cur = allx
    for _ in range(6):
        sm = sum(cur).to(torch.bfloat16).float()
        cur = [sm for _ in range(world)]
    assert torch.equal(static.cpu().float(), cur[0]), f"rank {rank} graph replay {rep}: max diff {(static.cpu().float() - cur[0]).abs().max()}"

# a tensor-parallel decode session over this communicator (the step `bench.py --gpus N` replays):
# logits of the vocabulary slices and greedy ids against the single-rank session of the same model
import numpy as np  # noqa: E402
cfg = decoder.ModelConfig("tp-proc-test", hidden=1024, layers=2, n_heads=8, n_kv=2, head_dim=128, inter=1024, vocab=4096)
spec = decoder.QuantSpec(4, 128)
ids0 = np.array([17, 923])
steps = 4


def run(model, comm_, graph):
    sess = decoder.DecodeSession(model, 2, max_len=32, span_len=16, kv_mode="none", comm=comm_)
    sess.set_state(ids0, [0, 0])
    if graph:
        sess.capture(warmup=1)
        sess.set_state(ids0, [0, 0])
    out = []
    for _ in range(steps):
        sess.replay() if graph else sess.step()
        torch.cuda.synchronize()
        out.append((sess.logits.cpu().numpy().copy(), sess.ids.cpu().numpy().copy()))
    return out


ref = run(decoder.build_random_model(cfg, spec, seed=99), None, False)
shard = decoder.build_random_model(cfg, spec, seed=99, rank=rank, nranks=world)
vloc = cfg.vocab // world
# (eager only: this harness gathers the arg-max pairs through gloo, which cannot be captured -- a node uses RCCL there)
for graph in (False,):
    dist.barrier()
    got = run(shard, comm, graph)
    for t in range(steps):
        want = ref[t][0][:, rank * vloc:(rank + 1) * vloc]
        err = float(np.abs(got[t][0] - want).max())
        assert err <= 1e-2, f"rank {rank} graph={graph} step {t}: logits differ by {err:.3e}"
        assert np.array_equal(got[t][1], ref[t][1]), f"rank {rank} graph={graph} step {t}: ids {got[t][1]} vs {ref[t][1]}"
dist.barrier()
print(f"P2P_PROC_OK rank {rank} backend {comm.backend}", flush=True)
