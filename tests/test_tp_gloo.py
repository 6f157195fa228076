"""Tensor-parallel path on CPU: world_size-2 `gloo` processes, each holding its tp.py shard of one
Qwen2-style decoder layer (weight-only quantised with the oracle quantiser on the FULL matrices,
then sliced -- the order the reference converter uses), computing with the numpy oracle and
exchanging with all_reduce exactly where the GPU decoder does (after o_proj and after down_proj,
residual carried by rank 0 only: gemm_op.cpp:133-137).  The sharded result must equal the
single-rank result to f32 summation-order noise.  Also: the partition tables for the BASELINE TP
degrees (Qwen2-7B g = 4 heads over 8 ranks -> KV replication, 18944 columns in units of 128)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import attention, gemm_ref, glue, quant
from oracle.numerics import bf16_round


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


CFG = dict(hidden=256, n=4, g=2, H=64, inter=512, group=128, L=9)


def _make_layer(seed=0):
    rng = np.random.default_rng(seed)
    c = CFG
    n, g, H = c["n"], c["g"], c["H"]
    w = {
        "qkv": rng.normal(0, 0.05, (c["hidden"], (n + 2 * g) * H)),
        "o": rng.normal(0, 0.05, (n * H, c["hidden"])),
        "gate": rng.normal(0, 0.05, (c["hidden"], c["inter"])),
        "up": rng.normal(0, 0.05, (c["hidden"], c["inter"])),
        "down": rng.normal(0, 0.05, (c["inter"], c["hidden"])),
    }
    q = {k: quant.iq_quantize_a16w8(bf16_round(v.astype(np.float32)), c["group"], "bf16") for k, v in w.items()}
    deq = {k: gemm_ref.dequant(*q[k], c["group"], 8) for k in q}  # what every rank's slice dequantises to
    x = {
        "h": rng.normal(0, 1, (1, c["hidden"])).astype(np.float32),
        "bias": bf16_round(rng.normal(0, 0.1, (n + 2 * g) * H).astype(np.float32)),
        "ln1": bf16_round(rng.normal(1, 0.1, c["hidden"]).astype(np.float32)),
        "ln2": bf16_round(rng.normal(1, 0.1, c["hidden"]).astype(np.float32)),
        "K": bf16_round(rng.normal(0, 1, (c["L"], g, H)).astype(np.float32)),
        "V": bf16_round(rng.normal(0, 1, (c["L"], g, H)).astype(np.float32)),
    }
    return deq, x


def _layer_forward(deq, x, shard, ffn_cols, rank, reduce_fn):
    """One decoder layer (no RoPE: position-independent check) on this rank's shard."""
    from dash_infer_amd import tp
    c = CFG
    n, g, H = c["n"], c["g"], c["H"]
    h = x["h"].copy()
    xn = bf16_round(glue.rmsnorm(h, x["ln1"], 1e-6))
    cols = tp.qkv_columns(shard, n, g, H)
    qkv = bf16_round(xn @ deq["qkv"][:, cols] + x["bias"][cols])
    nq, nk = len(shard.q_heads), len(shard.kv_heads)
    qh = qkv[0, : nq * H].reshape(nq, H)
    # local KV cache = the rank's KV heads of the (replicated) history
    K, V = x["K"][:, shard.kv_heads, :], x["V"][:, shard.kv_heads, :]
    hpg_local = nq // nk
    attn = np.concatenate([attention.decode_attention(qh[i * hpg_local:(i + 1) * hpg_local], K[:, i:i + 1], V[:, i:i + 1], 1.0 / np.sqrt(H))
                           for i in range(nk)])
    attn = bf16_round(attn.reshape(1, nq * H))
    part = attn @ deq["o"][tp.o_rows(shard, H), :]
    h = reduce_fn(part + (h if rank == 0 else 0.0))
    xn = bf16_round(glue.rmsnorm(h, x["ln2"], 1e-6))
    cols = list(ffn_cols)
    act = bf16_round(glue.silu(xn @ deq["gate"][:, cols]) * (xn @ deq["up"][:, cols]))
    part = act @ deq["down"][cols, :]
    return reduce_fn(part + (h if rank == 0 else 0.0))


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import _load_pkg
    _load_pkg()
    from dash_infer_amd import tp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    deq, x = _make_layer()
    c = CFG
    shard = tp.shard_heads(c["n"], c["g"], world)[rank]
    ffn = tp.shard_ffn(c["inter"], world, c["group"])[rank]

    def allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        dist.all_reduce(t)
        return t.numpy()

    out = _layer_forward(deq, x, shard, ffn, rank, allreduce)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_tp2_layer_equals_single_rank(pkg):
    from dash_infer_amd import tp
    deq, x = _make_layer()
    full = tp.shard_heads(CFG["n"], CFG["g"], 1)[0]
    ref = _layer_forward(deq, x, full, range(CFG["inter"]), 0, lambda a: a.astype(np.float32))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-5)


def test_partition_tables_baseline(pkg):
    from dash_infer_amd import tp
    # Qwen2-7B: n = 28, g = 4.  TP = 2, 4: whole KV heads per rank; TP = 8: every KV head on two ranks, its 7
    # query heads split 4 + 3 (the reference refuses this case: head_gqa.h:29-49)
    for nr in (1, 2, 4, 8):
        sh = tp.shard_heads(28, 4, nr)
        assert sorted(h for s in sh for h in s.q_heads) == list(range(28))
        for s in sh:
            assert all(h // 7 in s.kv_heads for h in s.q_heads)
    sh8 = tp.shard_heads(28, 4, 8)
    assert [len(s.q_heads) for s in sh8] == [4, 3] * 4 and [s.kv_heads for s in sh8] == [[k] for k in range(4) for _ in range(2)]
    with pytest.raises(ValueError):
        tp.shard_heads(28, 4, 3)
    # FFN columns: 18944 = 148 groups of 128 -> 8 ranks get 19 or 18 groups, contiguous, group aligned
    ffn = tp.shard_ffn(18944, 8, 128)
    assert [len(r) // 128 for r in ffn] == [19, 19, 19, 19, 18, 18, 18, 18] and ffn[0].start == 0 and ffn[-1].stop == 18944
    assert all(a.stop == b.start for a, b in zip(ffn, ffn[1:]))
    cols = tp.qkv_columns(sh8[1], 28, 4, 128)
    assert len(cols) == (3 + 2) * 128 and cols[0] == 4 * 128 and cols[3 * 128] == 28 * 128
