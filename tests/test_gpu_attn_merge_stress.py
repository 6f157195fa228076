"""Stress test of the decode attention's in-launch split merge: the merge hands the split partials from
workgroup to workgroup with write-through (sc1) stores, a drained vmcnt, a relaxed agent-scope ticket and sc1 loads -- no
release / acquire fences, i.e. it leans on gfx942 / gfx950 cache behaviour.  Here the decode-step attention (16-bit cache,
dihip_span_attn_decode_fused_sync) and the op-boundary kernels (int8 / uint4 caches, dihip_span_attn_decode_sync) run 200 times
back to back with many requests and many splits while a second stream saturates HBM and a third issues a stream of tiny kernels;
every repetition must be BIT-identical to the two-launch merge (no ticket words: partial records, then span_attn_split_merge_kernel)
computed on the idle GPU -- the arithmetic of the two forms is the same, so any difference is a stale or torn record."""
import numpy as np
import pytest
import torch

from tests.test_gpu_gemv_stress import Load, REPS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kv_mode,batch,n,g,seq", [("none", 1, 28, 4, 2048), ("none", 6, 28, 4, 1500), ("none", 32, 8, 1, 700),
                                                   ("i8", 4, 28, 4, 1800), ("u4", 8, 28, 4, 1200)])
def test_in_launch_merge_is_bit_identical_to_the_two_launch_merge_under_load(pkg, kv_mode, batch, n, g, seq):
    from dash_infer_amd import ops
    H, S = 128, 128
    max_len = seq + 8
    spr = (max_len + S - 1) // S
    gen = torch.Generator(device="cuda").manual_seed(seq + batch)
    pool = ops.SpanPool(2 * batch * spr + 1, g, S, H, kv_mode, torch.bfloat16)
    kv = ops.KVCacheSet(pool, batch, spr)
    for b in range(batch):
        kv.ensure(b, max_len)
    kv.sync()
    # random history
    if kv_mode == "none":
        v = pool.pool.view(torch.bfloat16)
        v.copy_(torch.randn(v.shape, generator=gen, device="cuda").to(torch.bfloat16))
    else:
        pool.pool.random_(0, 256, generator=gen)
        hb = H if kv_mode == "i8" else H // 2
        per = pool.aligned
        params = pool.pool.view(-1, per)[:, g * S * hb: g * S * hb + g * S * 8].contiguous().view(torch.float32).view(-1, g, S, 2)
        params[..., 0] = 8.0 if kv_mode == "u4" else 0.0
        params[..., 1] = 0.25 if kv_mode == "u4" else 0.02
        pool.pool.view(-1, per)[:, g * S * hb: g * S * hb + g * S * 8] = params.view(torch.uint8).view(-1, g * S * 8)
    lens_old = torch.tensor([seq - 1 - 37 * b % 200 for b in range(batch)], dtype=torch.int32, device="cuda")
    lens_new = lens_old + 1
    qkv = (torch.randn(batch, (n + 2 * g) * H, generator=gen, device="cuda") * 0.7).to(torch.bfloat16)
    ws = torch.empty(max(ops.span_attn_workspace(batch, n, H, max_len), ops.span_attn_fused_workspace(batch, n, g, H, max_len), 256),
                     dtype=torch.uint8, device="cuda")
    sync = torch.zeros(int(ops.lib().dihip_span_attn_sync_bytes(batch, n)), dtype=torch.uint8, device="cuda")
    scale = 1.0 / np.sqrt(H)
    out = torch.empty(batch, n * H, dtype=torch.bfloat16, device="cuda")
    if kv_mode == "none":
        inv_freq = (1.0 / (1e6 ** (torch.arange(0, H, 2, dtype=torch.float64) / H))).float().cuda()
        tab = ops.rope_table(inv_freq, max_len + 1, H)

        def run(with_ticket):
            ops.span_attn_decode_fused(qkv, kv, lens_old, tab, n, g, H, max_len, scale, ws, out=out, sync=sync if with_ticket else None)
    else:
        q = qkv[:, : n * H].contiguous()

        def run(with_ticket):
            ops.span_attn_decode(q, kv, lens_new, n, g, H, max_len, scale, ws, sync if with_ticket else None, out=out)

    run(False)                                   # two-launch merge on the idle GPU: the reference bits
    torch.cuda.synchronize()
    want = out.clone()
    run(True)
    torch.cuda.synchronize()
    assert torch.equal(out, want), "in-launch merge differs from the two-launch merge on an idle GPU"
    keep = torch.empty((REPS,) + tuple(want.shape), dtype=want.dtype, device="cuda")
    load = Load()
    load.enqueue()
    for r in range(REPS):
        out.zero_()
        run(True)
        keep[r].copy_(out, non_blocking=True)
    torch.cuda.synchronize()
    load.wait()
    same = (keep.view(REPS, -1).view(torch.uint8) == want.reshape(1, -1).view(torch.uint8)).all(dim=1)
    bad = (~same).nonzero().flatten().tolist()
    assert not bad, f"{len(bad)} of {REPS} repetitions differ from the two-launch merge (first: repetition {bad[0]})"
    assert int(sync.view(torch.int32).abs().sum()) == 0, "the ticket words were not left at zero"
