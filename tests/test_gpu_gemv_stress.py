"""Stress test of the hand-counted register ring of the decode GEMV.

Both wrong-result bugs of round 2 (row 0 of a 2-4-row call read before its load had landed; a tail dummy load landing on a
live register) were TIMING dependent: every parity test ran its shape once on an idle GPU and passed.  Here every
instantiation family of gemv_stream_kernel -- M = 1 ... 4, W4 (per-k-tile groups and wider groups) / W8 (per-channel and
sub-channel) / W16, the plain and RMSNorm prologues, the STD (bias + activation + residual) / SwiGLU / f32-hidden-stream
epilogues, and the mixture-of-experts slot forms (one slot per launch row, and groups of up to 4 slots) -- is launched
200 times back to back while a second stream saturates HBM with large copies and a third runs a stream of tiny kernels
(loaded memory latency, contended issue ports, waves of other kernels sharing the CUs: the conditions under which the
bugs showed).  Every repetition must be BIT-identical to the result computed on the idle GPU."""
import numpy as np
import pytest
import torch

from oracle import quant
from oracle.numerics import bf16_round

pytestmark = pytest.mark.gpu
REPS = 200


@pytest.fixture(scope="module")
def ops(pkg):
    from dash_infer_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


class Load:
    """background work on two side streams, enqueued ahead of the launches under test"""

    def __init__(self):
        self.s_bw, self.s_small = torch.cuda.Stream(), torch.cuda.Stream()
        self.a = torch.empty(192 * 1024 * 1024, dtype=torch.uint8, device="cuda")   # beyond L2, 3/4 of the Infinity Cache
        self.b = torch.empty_like(self.a)
        self.t = torch.zeros(4096, device="cuda")

    def enqueue(self, copies=24, smalls=600):
        with torch.cuda.stream(self.s_bw):
            for _ in range(copies):          # ~0.1 ms each at a few TB/s: covers the 200 launches under test
                self.b.copy_(self.a, non_blocking=True)
        with torch.cuda.stream(self.s_small):
            for _ in range(smalls):
                self.t.add_(1.0)

    def wait(self):
        self.s_bw.synchronize()
        self.s_small.synchronize()


def dev(a, dt=torch.bfloat16):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dt).cuda()


def packed(ops, rng, K, N, wbits, G):
    W = bf16_round(rng.normal(0, 0.02, (K, N)).astype(np.float32))
    q, s, z = (quant.iq_quantize_a16w8 if wbits == 8 else quant.iq_quantize_a16w4)(W, G, "bf16")
    return ops.pack_lowp(torch.from_numpy(q).cuda(), dev(s), dev(z), G, wbits)


def stress(load, launch, out_of):
    """launch(): one launch of the call under test into its output; out_of(): that output tensor.  Idle result first, then
    REPS launches under load, each copied aside on the launch stream; all must equal the idle bits."""
    launch()
    torch.cuda.synchronize()
    want = out_of().clone()
    keep = torch.empty((REPS,) + tuple(want.shape), dtype=want.dtype, device="cuda")
    load.enqueue()
    for r in range(REPS):
        launch()
        keep[r].copy_(out_of(), non_blocking=True)
    torch.cuda.synchronize()
    load.wait()
    same = (keep.view(REPS, -1).view(torch.uint8) == want.reshape(1, -1).view(torch.uint8)).all(dim=1)
    bad = (~same).nonzero().flatten().tolist()
    assert not bad, f"{len(bad)} of {REPS} repetitions differ from the idle result (first: repetition {bad[0]})"


@pytest.mark.parametrize("wbits,G", [(4, 128), (4, 256), (8, -1), (8, 64), (8, 128)])
@pytest.mark.parametrize("M", [1, 2, 3, 4])
def test_gemv_ring_bit_stable_under_load(ops, wbits, G, M):
    rng = np.random.default_rng(100 * wbits + M + abs(G))
    K, I, Nq = 3584, 4736, 1152        # decode-layer proportions (a quarter of the 7B intermediate width): seconds per case
    load = Load()
    h = torch.from_numpy(rng.normal(0, 1.5, (M, K)).astype(np.float32)).cuda()
    gamma = dev(rng.normal(1, 0.1, K))
    sc = ops.Scratch(max(ops.lowp_workspace_bytes(wbits, M, I, K, G), ops.lowp_workspace_bytes(wbits, M, K, I, G),
                         ops.lowp_workspace_bytes(wbits, M, Nq, K, G)))
    # RMSNorm prologue + STD epilogue (bias): the qkv projection
    pq = packed(ops, rng, K, Nq, wbits, G)
    bias = dev(rng.normal(0, 0.3, Nq))
    y = torch.empty(M, Nq, dtype=torch.bfloat16, device="cuda")
    stress(load, lambda: ops.fused_norm_gemm(h, gamma, 1e-6, pq, bias, sc, out=y), lambda: y)
    # RMSNorm prologue + SwiGLU epilogue over a gate / up pair
    pg, pu = packed(ops, rng, K, I, wbits, G), packed(ops, rng, K, I, wbits, G)
    act = torch.empty(M, I, dtype=torch.bfloat16, device="cuda")
    stress(load, lambda: ops.fused_norm_swiglu(h, gamma, 1e-6, pg, pu, sc, out=act), lambda: act)
    # plain prologue + f32 hidden-stream epilogue (residual): the down projection
    pd = packed(ops, rng, I, K, wbits, G)
    hout = torch.empty(M, K, dtype=torch.float32, device="cuda")
    stress(load, lambda: ops.fused_gemm_addto(act, pd, h, sc, out=hout), lambda: hout)
    # plain prologue + STD epilogue with activation and FT residual (op-boundary GemmA16Wx)
    res = dev(rng.normal(0, 1, (M, K)))
    y2 = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    stress(load, lambda: ops.gemm_lowp(act, pd, residual=res, act="gelu_tanh", scratch=sc, out=y2), lambda: y2)


@pytest.mark.parametrize("M", [1, 3])
def test_dense_gemv_ring_bit_stable_under_load(ops, M):
    """W16 (lm_head / router): RMSNorm prologue, f32 output"""
    rng = np.random.default_rng(16 + M)
    K, V = 3584, 19008
    load = Load()
    h = torch.from_numpy(rng.normal(0, 1.5, (M, K)).astype(np.float32)).cuda()
    gamma = dev(rng.normal(1, 0.1, K))
    pw = ops.pack_dense(dev(rng.normal(0, 0.02, (K, V))))
    sc = ops.Scratch(int(ops.lib().dihip_dense_workspace_bytes(M, V, K)))
    logits = torch.empty(M, V, dtype=torch.float32, device="cuda")
    stress(load, lambda: ops.lm_head(h, gamma, 1e-6, pw, sc, out=logits), lambda: logits)


@pytest.mark.parametrize("wbits,G,T", [(8, -1, 1), (8, -1, 6), (4, 128, 5)])
def test_moe_slot_gemv_bit_stable_under_load(ops, wbits, G, T):
    """the SLOT forms: one slot per launch row (T = 1) and groups of up to 4 slots that picked the same expert (T > 1)"""
    rng = np.random.default_rng(wbits + T)
    hidden, proj, E, top_k = 1024, 512, 6, 2
    load = Load()
    def stack(K, N):
        qs, ss, zs = [], [], []
        for _ in range(E):
            W = bf16_round(rng.normal(0, 0.05, (K, N)).astype(np.float32))
            q, s, z = (quant.iq_quantize_a16w8 if wbits == 8 else quant.iq_quantize_a16w4)(W, G, "bf16")
            qs.append(torch.from_numpy(q).cuda()); ss.append(dev(s)); zs.append(dev(z))
        return ops.pack_experts(qs, ss, zs, G, wbits)
    gate, up, down = stack(hidden, proj), stack(hidden, proj), stack(proj, hidden)
    x = dev(rng.normal(0, 1, (T, hidden)))
    experts = torch.from_numpy(np.stack([rng.permutation(3)[:top_k] for _ in range(T)]).astype(np.int32)).cuda()  # 3 experts: groups form
    scores = torch.from_numpy(rng.uniform(0.1, 0.6, (T, top_k)).astype(np.float32)).cuda()
    ws = torch.empty(int(ops.lib().dihip_moe_workspace_bytes(T, top_k, hidden, proj)), dtype=torch.uint8, device="cuda")
    out = torch.empty(T, hidden, dtype=torch.bfloat16, device="cuda")
    stress(load, lambda: ops.moe_experts(x, experts, scores, gate, up, down, ws=ws, out=out), lambda: out)


def test_decode_attention_in_launch_merge_bit_stable_under_load(ops):
    """the in-launch merge of the decode-step attention hands records between workgroups (write-through stores, arrival
    ticket, L1-bypassing loads): under load and with the consumer's caches warm from the previous repetition, every one of
    200 launches must reproduce the two-launch result -- a stale record would show as a differing row"""
    from tests.test_gpu_kv_attn import build_batch
    rng = np.random.default_rng(5)
    n, g, H, S = 28, 4, 128, 128
    lens = [2047]
    B = 1
    pool, kv, _, _ = build_batch(ops, rng, lens, n, g, H, S, "none", "bf16", extra_tokens=1)
    load = Load()
    max_len = 2048 + 64
    tab = ops.rope_table(torch.from_numpy((1.0 / (1e6 ** (np.arange(0, H, 2) / H))).astype(np.float32)).cuda(), max_len + 1, H)
    ws = torch.empty(ops.span_attn_fused_workspace(B, n, g, H, max_len), dtype=torch.uint8, device="cuda")
    tickets = torch.zeros(int(ops.lib().dihip_span_attn_sync_bytes(B, n)), dtype=torch.uint8, device="cuda")
    old = torch.tensor(lens, dtype=torch.int32, device="cuda")
    out = torch.empty(B, n * H, dtype=torch.bfloat16, device="cuda")
    qkvs = [dev(rng.normal(0, 1, (B, (n + 2 * g) * H))) for _ in range(2)]
    # two different inputs alternate, so that a record left over from the previous launch is a WRONG record
    state = {"i": 0}
    def launch_for(which, sync):
        return lambda: ops.span_attn_decode_fused(qkvs[which], kv, old, tab, n, g, H, max_len, 1.0 / np.sqrt(H), ws, out=out, sync=sync)
    wants = []
    for w in range(2):
        launch_for(w, None)()
        torch.cuda.synchronize()
        wants.append(out.clone())
    keep = torch.empty((REPS, B, n * H), dtype=torch.bfloat16, device="cuda")
    load.enqueue()
    for r in range(REPS):
        launch_for(r & 1, tickets)()
        keep[r].copy_(out, non_blocking=True)
    torch.cuda.synchronize()
    load.wait()
    for r in range(REPS):
        assert torch.equal(keep[r], wants[r & 1]), f"repetition {r}: the in-launch merge differs from the two-launch result"
    assert int(tickets.view(torch.int32).abs().sum()) == 0
