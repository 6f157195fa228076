"""The guarded set-up of the peer-to-peer all-reduce (decoder.P2PComm(guarded=True), make_comm(backend="auto")) between two
`gloo` processes on CPU, against a FAKE device library: what is under test is the control flow between the ranks -- a stage
that fails on ONE rank (opening a peer's IPC handle, the probe exchange timing out) must end in the same decision on EVERY
rank (RCCL, labelled), with no rank left alone in a collective.  The device side of the same path runs in
tests/test_gpu_p2p_processes.py."""
import ctypes as C
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class FakeLib:
    """the handful of C-ABI entries P2PComm calls; `fail` = (stage, rank) that reports an error"""

    def __init__(self, rank, fail):
        self.rank, self.fail = rank, fail

    def _bad(self, stage):
        return self.fail is not None and self.fail == (stage, self.rank)

    def dihip_p2p_ar_alloc(self, buf):
        buf._obj.value = 0x1000 + self.rank
        return 4 if self._bad("alloc") else 0

    def dihip_ipc_get_handle(self, buf, raw):
        raw[0] = self.rank + 1
        return 0

    def dihip_ipc_open_handle(self, handle, p):
        p._obj.value = 0x2000 + handle[0]
        return 5 if self._bad("open") else 0

    def dihip_p2p_ar_create(self, handle, rank, nranks, ptrs):
        handle._obj.value = 0x3000 + rank
        return 0

    def dihip_p2p_ar_max_bytes(self):
        return 256 * 1024

    def dihip_p2p_ar_set_timeout(self, handle, spins, trap):
        return 0

    def dihip_p2p_ar_error(self, handle, err):
        err._obj.value = 1 if self._bad("probe") else 0
        return 0


class FakeRccl:
    backend = "rccl"

    def __init__(self, rank, nranks, device):
        self.rank, self.nranks = rank, nranks

    def allreduce_(self, t):
        dist.all_reduce(t)
        return t

    def allgather(self, src, dst):
        dist.all_gather_into_tensor(dst, src)
        return dst


def _worker(rank, world, port, fail, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import _load_pkg
    _load_pkg()
    from dash_infer_amd import decoder
    fake = FakeLib(rank, fail)
    decoder.lib = lambda: fake
    decoder.RcclComm = FakeRccl
    if fail is not None and fail[0] == "rccl":   # the C-ABI RCCL communicator cannot be created (on every rank)
        def broken(*a, **k):
            raise RuntimeError("ncclCommInitRank failed")
        decoder.RcclComm = broken
    # the fake "device" all-reduce involves no host collective (like the real kernels: a rank that left the probe early does
    # not pair up with the others' later calls -- theirs time out in probe mode); every tensor here is full of rank + 1
    decoder.P2PComm.allreduce_ = lambda self, t: t.fill_(world * (world + 1) / 2)
    comm = decoder.make_comm(rank, world, torch.device("cpu"), backend="auto", allow_labelled_fallback=True)
    t = torch.full((8,), float(rank + 1))
    comm.allreduce_(t)                      # whatever was chosen, the ranks still talk to one another
    q.put((rank, comm.backend, float(t[0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail,expect", [(None, "p2p-oneshot"), (("open", 1), "rccl (p2p-oneshot unavailable"),
                                         (("probe", 0), "rccl (p2p-oneshot unavailable"), (("alloc", 1), "rccl (p2p-oneshot unavailable"),
                                         (("rccl", -1), "p2p-oneshot (all-gather / long messages: torch.distributed")])
def test_guarded_p2p_setup_ends_in_the_same_backend_on_every_rank(pkg, fail, expect):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fail, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [0, 1]
    assert all(g[1].startswith(expect) for g in got), got
    assert got[0][1] == got[1][1] or expect.startswith("rccl")   # the same decision (the message names the failing rank's view)
    assert all(g[2] == 3.0 for g in got)


def _verify_worker(rank, world, port, corrupt_rank, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import _load_pkg
    _load_pkg()
    from dash_infer_amd import decoder
    decoder.lib = lambda: FakeLib(rank, None)
    decoder.RcclComm = FakeRccl
    calls = {"n": 0}

    def device_sum(self, t):   # the "peer-to-peer kernel": a correct sum, except for one stale word on one rank every third call
        dist.all_reduce(t)
        calls["n"] += 1
        if rank == corrupt_rank and calls["n"] % 3 == 0:
            t.view(-1)[5] += 0.75
    decoder.P2PComm._device_sum = device_sum
    comm = decoder.make_comm(rank, world, torch.device("cpu"), backend="auto", allow_labelled_fallback=True)
    assert comm.backend.startswith("p2p-oneshot")
    comm.start_verification()
    for i in range(7):
        comm.allreduce_(torch.full((3584,), float(rank + 1) * (i + 1), dtype=torch.bfloat16 if i % 2 else torch.float32))
    ok, summary = comm.finish_verification(torch.device("cpu"))
    q.put((rank, ok, summary))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt_rank", [-1, 1])
def test_p2p_verification_phase_reaches_one_verdict_on_every_rank(pkg, corrupt_rank):
    """bench.py's verification of the peer-to-peer all-reduce (every sum of two eager decode steps checked against RCCL): a
    stale word seen by ONE rank makes EVERY rank report failure (and fall back together); clean sums pass on both."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verify_worker, args=(r, 2, port, corrupt_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [corrupt_rank < 0] * 2, got
    assert all("7 all-reduces checked" in g[2] for g in got), got
