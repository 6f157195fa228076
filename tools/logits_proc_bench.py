"""Times dihip_logits_processor / dihip_logprobs (csrc/logits_proc.hip) at the Qwen2 vocabulary: graph-replayed launches between events.
    python tools/logits_proc_bench.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g._load_pkg()
from dash_infer_amd import ops  # noqa: E402
from dash_infer_amd.capi import check, lib  # noqa: E402


def timed(fn, iters=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(10):
                fn()
        gr.replay()
        s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(iters // 10):
            gr.replay()
        b.record(s)
        s.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    N = 152064
    rng = np.random.default_rng(0)
    for M, L in [(1, 2048), (1, 8192), (32, 2048)]:
        logits = torch.from_numpy(rng.normal(0, 3, (M, N)).astype(np.float32)).cuda()
        ids = torch.from_numpy(rng.integers(0, N, (M, L)).astype(np.int64)).cuda()
        i32 = lambda v: torch.full((M,), v, dtype=torch.int32, device="cuda")
        f32 = lambda v: torch.full((M,), v, dtype=torch.float32, device="cuda")
        cur, inp, rep, fq, pr, ng, ml, eos, sup = i32(L), i32(L // 2), f32(1.2), f32(0.1), f32(0.3), i32(3), i32(0), i32(5), i32(0)
        ws = torch.empty(M * N * 4, dtype=torch.uint8, device="cuda")
        chosen = torch.zeros(M, dtype=torch.int64, device="cuda")
        tok = torch.empty(M, dtype=torch.float32, device="cuda")
        tv = torch.empty(M, 10, dtype=torch.float32, device="cuda")
        ti = torch.empty(M, 10, dtype=torch.int32, device="cuda")
        p = ops.ptr

        def proc():
            check(lib().dihip_logits_processor(ops.cur_stream(), p(logits), M, N, p(ids), L, p(cur), p(inp), p(rep), p(fq), p(pr), p(ng), p(ml), p(eos),
                                               p(sup), p(ws), ws.numel()))

        lws = torch.empty(int(lib().dihip_logprobs_workspace_bytes(M, N, 10)) + 8, dtype=torch.uint8, device="cuda")

        def lprob(top):
            check(lib().dihip_logprobs(ops.cur_stream(), p(logits), M, N, p(chosen), top, 10, p(tok), p(tv), p(ti), p(lws), lws.numel()))

        print(f"M={M:3d} vocab={N} history={L:5d}: logits_processor {timed(proc):7.2f} us   logprobs top-0 {timed(lambda: lprob(0)):7.2f} us   "
              f"top-5 {timed(lambda: lprob(5)):7.2f} us   top-10 {timed(lambda: lprob(10)):7.2f} us", flush=True)


if __name__ == "__main__":
    main()
