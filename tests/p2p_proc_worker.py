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
dist.barrier()
print(f"P2P_PROC_OK rank {rank} backend {comm.backend}", flush=True)
