"""bench.py's timing contract on CPU: world-size-2 `gloo` processes run bench.timed_blocks over a fake step whose
duration depends on the rank.  Every rank must report the SAME block times, equal to the SLOWEST rank's (max over ranks), the
reported time must be the median block, and a JSON line built from it carries n_gpus = world size and value = whole-job
steps / that time."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    per_step = [0.004, 0.012][rank]          # rank 1 is three times slower
    calls = {"n": 0, "before": 0}

    def run_n(n):
        calls["n"] += n
        # one block (the third) is disturbed on rank 0 only: the median must not move
        extra = 0.2 if (rank == 0 and calls["before"] == 3) else 0.0
        time.sleep(n * per_step + extra)

    def before():
        calls["before"] += 1

    med, times = bench.timed_blocks(run_n, 5, 5, world=world, device=torch.device("cpu"), sync=lambda: None, before_block=before)
    out[rank] = (med, times, calls["n"], calls["before"], bench.blocks_summary(times, 5))
    dist.barrier()
    dist.destroy_process_group()


def test_block_times_are_the_max_over_ranks_and_the_median_is_reported():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (m0, t0, n0, b0, s0), (m1, t1, n1, b1, s1) = out[0], out[1]
    assert n0 == n1 == 25 and b0 == b1 == 5                    # exactly K steps per block, `blocks` blocks, on every rank
    assert t0 == pytest.approx(t1, abs=1e-9) and m0 == pytest.approx(m1, abs=1e-9)   # one verdict on all ranks
    slow = 5 * 0.012
    assert all(t >= slow * 0.98 for t in t0)                   # never the fast rank's time
    assert max(t0) >= 0.2                                      # the disturbed block is in the list ...
    assert slow * 0.98 <= m0 <= slow * 1.6                     # ... and not in the reported time
    assert s0["count"] == 5 and s0["steps_each"] == 5 and s0["ms_per_step_max"] > s0["ms_per_step_median"] >= s0["ms_per_step_min"]
    # the line: whole-job tokens / max-over-ranks time, n_gpus = world
    batch, steps = 1, 5
    line = {"n_gpus": world, "value": batch * steps / m0, "ms_per_step": m0 / steps * 1e3}
    assert line["n_gpus"] == 2 and line["value"] == pytest.approx(steps / m0)
