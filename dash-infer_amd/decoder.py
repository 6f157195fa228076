"""Decode-step runner for a Qwen2-style decoder on the dihip C-ABI (bench / smoke / parity tests).

This is NOT the reference's engine (AsEngine / AsModel are out of scope): it is the smallest
graph executor that strings the hot-path ops of one decoder step together in the order of
python/pyhie/allspark/model/qwen_v15.py:210-388, so that decode tokens/s can be measured and
greedy token ids compared with the oracle:

  embedding -> L x [ RMSNorm+qkv GEMV(+bias) -> RoPE+KV append -> SpanAttention
                     -> o_proj GEMV (+residual, [all-reduce]) -> RMSNorm+gate/up GEMV+SwiGLU
                     -> down GEMV (+residual, [all-reduce]) ]
            -> final RMSNorm + lm_head -> greedy argmax -> step counters += 1

Every arithmetic step is a call into libdashinfer_hip.so; torch provides device memory, streams,
graph capture (hipGraph) and the process group used to bootstrap RCCL.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import os

import torch

from . import capi, ops, quantize, tp
from .capi import check, lib


@dataclass
class MoEConfig:
    """Mixture-of-experts feed-forward block (python/pyhie/allspark/model/qwen_v20_moe.py:318-382): `num_experts` routed experts
    (hidden -> moe_inter -> hidden), `top_k` per token, + one shared expert of width ModelConfig.inter behind a sigmoid gate."""
    num_experts: int
    top_k: int
    moe_inter: int


@dataclass
class ModelConfig:
    name: str
    hidden: int
    layers: int
    n_heads: int
    n_kv: int
    head_dim: int
    inter: int
    vocab: int
    eps: float = 1e-6
    rope_theta: float = 1000000.0
    moe: Optional[MoEConfig] = None   # set: every layer's feed-forward block is the MoE block (inter = the shared expert's width)


QWEN2_7B = ModelConfig("Qwen2-7B", 3584, 28, 28, 4, 128, 18944, 152064)
QWEN2_72B = ModelConfig("Qwen2-72B", 8192, 80, 64, 8, 128, 29696, 152064)  # GPTQ-padded intermediate
# BASELINE configs[4]: 64 experts of width 2560, top-8, shared expert of width 20480
QWEN2_57B_A14B = ModelConfig("Qwen2-57B-A14B", 3584, 28, 28, 4, 128, 20480, 151936, moe=MoEConfig(64, 8, 2560))


@dataclass
class QuantSpec:
    wbits: int            # 4 | 8
    group: int            # -1 per-channel | 64/128/...
    gptq_like_zeros: bool = False


@dataclass
class LayerWeights:
    ln1: torch.Tensor
    qkv: ops.PackedWeight
    qkv_bias: torch.Tensor
    o: ops.PackedWeight
    ln2: torch.Tensor
    gate: ops.PackedWeight
    up: ops.PackedWeight
    down: ops.PackedWeight
    # mixture-of-experts layers (cfg.moe): gate / up / down above are the SHARED expert
    router: Optional[ops.PackedWeight] = None        # W16 [hidden, num_experts] (replicated)
    shared_sig: Optional[ops.PackedWeight] = None    # W16 [hidden, 1]: the shared expert's sigmoid gate (replicated)
    exp_gate: Optional[ops.PackedExperts] = None     # this rank's experts (expert parallelism: E / nranks consecutive experts)
    exp_up: Optional[ops.PackedExperts] = None
    exp_down: Optional[ops.PackedExperts] = None
    ep: Optional[tuple] = None                       # (first expert, count) of this rank
    expert_bytes: int = 0                            # packed bytes of ONE routed expert (gate + up + down, with scales / zeros)


@dataclass
class ModelWeights:
    cfg: ModelConfig
    quant: QuantSpec
    rank: int
    nranks: int
    heads: tp.HeadShard
    embed: torch.Tensor           # FT [vocab, hidden] (replicated)
    layers: List[LayerWeights]
    final_norm: torch.Tensor
    lm_head: ops.PackedWeight     # W16, this rank's vocabulary slice
    vocab_offset: int
    vocab_local: int
    weight_bytes: int = 0         # packed weight + (scale, zero) bytes streamed per decode step


def _pack(q, s, z, spec, rows=None, cols=None, n_full=None):
    """Slice a quantised [K,N] weight (rows / cols = index lists or None) and pack it for the GPU."""
    return ops.pack_lowp(*_slice(q, s, z, spec, rows, cols, n_full), spec.group, spec.wbits)


def _slice(q, s, z, spec, rows=None, cols=None, n_full=None):
    """A rank's slice of a quantised [K,N] weight, unpacked as the converter would hand it to the operator: (q, scales, zeros)."""
    if rows is None and cols is None:
        return q.contiguous(), s.contiguous(), z.contiguous()
    if spec.wbits == 4:
        # work on unpacked nibbles for column slicing
        K = q.shape[0]
        un = torch.empty(K, q.shape[1] * 2, dtype=torch.uint8, device=q.device)
        un[:, 0::2] = q & 0xF
        un[:, 1::2] = q >> 4
        un = un[:, :n_full]
        if cols is not None:
            un = un[:, cols]
        if rows is not None:
            un = un[rows, :]
        if un.shape[1] % 2:
            un = torch.nn.functional.pad(un, (0, 1))
        qq = ((un[:, 1::2] << 4) | (un[:, 0::2] & 0xF)).contiguous()
    else:
        qq = q
        if cols is not None:
            qq = qq[:, cols]
        if rows is not None:
            qq = qq[rows, :]
        qq = qq.contiguous()
    ss, zz = s, z
    if cols is not None:
        ss, zz = ss[:, cols], zz[:, cols]
    if rows is not None and spec.group > 0:
        # sub-channel parameters follow the K split (whole groups per rank, qwen_v15.py:540-569)
        g = spec.group
        assert len(rows) % g == 0 and rows[0] % g == 0, "row split must be group aligned"
        grp = torch.tensor([r // g for r in rows[::g]], device=s.device)
        ss, zz = ss[grp, :], zz[grp, :]
    return qq, ss.contiguous(), zz.contiguous()


def decisive_permutation(vocab, seed):
    """The next-token map of a `decisive` synthetic model: a fixed random permutation of the vocabulary (host generator:
    the same on every box and device)."""
    g = torch.Generator()
    g.manual_seed(int(seed) + 900004)
    return torch.randperm(vocab, generator=g)


def build_random_model(cfg: ModelConfig, spec: QuantSpec, seed=1234, device="cuda", rank=0, nranks=1,
                       dtype=torch.bfloat16, layers: Optional[int] = None, keep_fp=False, lm_head_split=None,
                       decisive: Optional[float] = None):
    """Synthetic weights per SURVEY 8(d): W ~ N(0, 0.02^2) in FT with
    Generator(seed + 1000*layer + idx), InstantQuant-quantised on the FULL matrix (as the reference
    converter does before the TP split), then sliced for this rank and packed.

    decisive = s (tests only): a model whose greedy choice is DECISIVE, like a trained model's and unlike
    N(0, 0.02) weights (whose top-2 logit margins of 1e-3 .. 6e-2 sit below any bf16 graph's own rounding noise, so that
    "bit-exact greedy ids" cannot be asserted on them).  The embedding rows are drawn with standard deviation s (the 28 random
    layers add ~ N(0, 2.3^2) per element each to the residual stream, so s = 2 leaves the embedding ~ 3 % of the final hidden
    state's energy) and the lm_head column of token pi(t) is 0.02 x the embedding row of t, pi = decisive_permutation: the
    logit of pi(current token) stands ~ 2 x above the best of the other 152k -- a margin tens of times the accumulated
    rounding error of the hidden state -- while every layer's output still moves every logit.  Everything else is unchanged."""
    H, n, g = cfg.head_dim, cfg.n_heads, cfg.n_kv
    shards = tp.shard_heads(n, g, nranks)
    me = shards[rank]
    ffn = tp.shard_ffn(cfg.inter, nranks, max(spec.group, 128) if spec.group > 0 else 128)[rank]
    ffn_cols = list(ffn)
    gen = torch.Generator(device=device)
    fp = {} if keep_fp else None

    def rand(shape, s, std=0.02):
        gen.manual_seed(s)
        return (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * std).to(dtype)

    L = cfg.layers if layers is None else layers
    out_layers, wbytes = [], 0
    for li in range(L):
        base = seed + 1000 * li
        w_qkv = rand((cfg.hidden, (n + 2 * g) * H), base + 0)
        b_qkv = rand(((n + 2 * g) * H,), base + 1)
        w_o = rand((n * H, cfg.hidden), base + 2)
        w_gate = rand((cfg.hidden, cfg.inter), base + 3)
        w_up = rand((cfg.hidden, cfg.inter), base + 4)
        w_down = rand((cfg.inter, cfg.hidden), base + 5)
        ln1 = (1.0 + rand((cfg.hidden,), base + 6, 0.1).float()).to(dtype)
        ln2 = (1.0 + rand((cfg.hidden,), base + 7, 0.1).float()).to(dtype)
        qs = {}
        for name, w in (("qkv", w_qkv), ("o", w_o), ("gate", w_gate), ("up", w_up), ("down", w_down)):
            qs[name] = quantize.quantize(w, spec.wbits, spec.group, spec.gptq_like_zeros)
        if fp is not None:
            fp[li] = {"qkv": qs["qkv"], "o": qs["o"], "gate": qs["gate"], "up": qs["up"], "down": qs["down"],
                      "qkv_bias": b_qkv, "ln1": ln1, "ln2": ln2}
        cols = tp.qkv_columns(me, n, g, H) if nranks > 1 else None
        rows_o = tp.o_rows(me, H) if nranks > 1 else None
        if fp is not None and nranks > 1:
            # under tensor parallelism fp holds THIS RANK'S slices, as the reference's converter splits them for the operators
            # (GROUP_VSPLIT qkv, HSPLIT o / down, VSPLIT gate / up; qwen_v15.py:540-569): what ref_graph.register_weights binds
            fp[li].update(qkv=_slice(*qs["qkv"], spec, cols=cols, n_full=(n + 2 * g) * H), qkv_bias=b_qkv[cols].contiguous(),
                          o=_slice(*qs["o"], spec, rows=rows_o, n_full=cfg.hidden),
                          gate=_slice(*qs["gate"], spec, cols=ffn_cols, n_full=cfg.inter), up=_slice(*qs["up"], spec, cols=ffn_cols, n_full=cfg.inter),
                          down=_slice(*qs["down"], spec, rows=ffn_cols, n_full=cfg.hidden))
        lw = LayerWeights(
            ln1=ln1,
            qkv=_pack(*qs["qkv"], spec, cols=cols, n_full=(n + 2 * g) * H),
            qkv_bias=(b_qkv[cols] if cols is not None else b_qkv).contiguous(),
            o=_pack(*qs["o"], spec, rows=rows_o, n_full=cfg.hidden),
            ln2=ln2,
            gate=_pack(*qs["gate"], spec, cols=ffn_cols if nranks > 1 else None, n_full=cfg.inter),
            up=_pack(*qs["up"], spec, cols=ffn_cols if nranks > 1 else None, n_full=cfg.inter),
            down=_pack(*qs["down"], spec, rows=ffn_cols if nranks > 1 else None, n_full=cfg.hidden),
        )
        wbytes += sum(p.nbytes for p in (lw.qkv, lw.o, lw.gate, lw.up, lw.down))
        if cfg.moe is not None:
            mc = cfg.moe
            assert mc.num_experts % nranks == 0, "expert parallelism: the experts must divide over the ranks (moe_op.cpp:103-105)"
            per = mc.num_experts // nranks
            lw.ep = (rank * per, per)
            w_router = rand((cfg.hidden, mc.num_experts), base + 8, std=0.5)   # spread-out logits: a decisive top-k
            w_sig = rand((cfg.hidden, 1), base + 9, std=0.05)
            lw.router, lw.shared_sig = ops.pack_dense(w_router), ops.pack_dense(w_sig)
            eq = {"gate": [], "up": [], "down": []}
            for e in range(mc.num_experts):
                mine = lw.ep[0] <= e < lw.ep[0] + per
                if not (mine or fp is not None):
                    continue
                for j, (name, shape) in enumerate((("gate", (cfg.hidden, mc.moe_inter)), ("up", (cfg.hidden, mc.moe_inter)),
                                                   ("down", (mc.moe_inter, cfg.hidden)))):
                    eq[name].append((e, quantize.quantize(rand(shape, base + 20 + 3 * e + j), spec.wbits, spec.group, spec.gptq_like_zeros)))
            def stack(name):
                local = [q for e, q in eq[name] if lw.ep[0] <= e < lw.ep[0] + per]
                return ops.pack_experts([q[0] for q in local], [q[1] for q in local], [q[2] for q in local], spec.group, spec.wbits)
            lw.exp_gate, lw.exp_up, lw.exp_down = stack("gate"), stack("up"), stack("down")
            # streamed per token: the router, top_k experts, the shared expert (already counted) and its gate
            per_expert = (lw.exp_gate.w.numel() + lw.exp_gate.sz.numel() + lw.exp_up.w.numel() + lw.exp_up.sz.numel()
                          + lw.exp_down.w.numel() + lw.exp_down.sz.numel()) // per
            wbytes += lw.router.nbytes + lw.shared_sig.nbytes   # the routed experts a step touches depend on the routing:
            lw.expert_bytes = per_expert                         # DecodeSession.count_distinct_experts() measures them
            if fp is not None:
                fp[li]["moe"] = {"router": w_router, "shared_gate_w": w_sig, "top_k": mc.top_k,
                                 "experts_gate": [q for _, q in eq["gate"]], "experts_up": [q for _, q in eq["up"]],
                                 "experts_down": [q for _, q in eq["down"]]}
        out_layers.append(lw)
        del w_qkv, w_o, w_gate, w_up, w_down, qs
    embed = rand((cfg.vocab, cfg.hidden), seed + 900001, std=0.02 if decisive is None else float(decisive))
    final_norm = (1.0 + rand((cfg.hidden,), seed + 900002, 0.1).float()).to(dtype)
    # lm_head stays unquantised FT (qwen_v15.py:153-164).  Under TP, two splits (lm_head_split, default $DIHIP_TP_LMHEAD or "vocab"):
    #   "vocab"  vocabulary-parallel slice + arg-max pair all-gather: 1/nranks of the logits per rank, 8 bytes per request on the
    #            wire -- right for greedy decoding, but no rank ever holds the full logits row;
    #   "k"      the reference's own graph (model_base.py:690-703: Gemm with attribute splitk, lm_head.weight HSPLIT, then
    #            AllReduce "all_reduce_lmhead"; gemm_op.cpp:95-98: lda = k * nranks): every rank multiplies its K slice of the
    #            normalised row with its row block of the weight, the partial logits are summed by an all-reduce, and every rank
    #            holds the FULL f32 logits row (top-k / top-p sampling, logprobs, the logits parity check under TP).
    lm_split = (lm_head_split or os.environ.get("DIHIP_TP_LMHEAD", "vocab")) if nranks > 1 else "vocab"
    assert lm_split in ("vocab", "k"), f"unknown lm_head split {lm_split!r}"
    vloc = cfg.vocab // nranks
    assert cfg.vocab % nranks == 0
    if decisive is None:
        w_lm = rand((cfg.hidden, cfg.vocab), seed + 900003)
    else:
        perm = decisive_permutation(cfg.vocab, seed).to(device)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(cfg.vocab, device=device)
        # column v = 0.02 x the embedding row of the token whose successor v is (FT rounding of the product)
        w_lm = (embed.float() * 0.02).to(dtype)[inv].t().contiguous()
        del perm, inv
    if fp is not None:
        fp["embed"], fp["final_norm"], fp["lm_head"] = embed, final_norm, w_lm
    if lm_split == "k":
        assert cfg.hidden % (32 * nranks) == 0, "K-split lm_head: hidden must split into whole 32-row k-tiles per rank"
        kloc = cfg.hidden // nranks
        lm = ops.pack_dense(w_lm[rank * kloc:(rank + 1) * kloc, :].contiguous())
        if fp is not None and nranks > 1:
            fp["lm_head"] = w_lm[rank * kloc:(rank + 1) * kloc, :].contiguous()   # HSPLIT (model_base.py:690-703)
    else:
        lm = ops.pack_dense(w_lm[:, rank * vloc:(rank + 1) * vloc].contiguous())
    wbytes += lm.nbytes
    del w_lm
    torch.cuda.synchronize()
    if lm_split == "k":
        mw = ModelWeights(cfg, spec, rank, nranks, me, embed, out_layers, final_norm, lm, 0, cfg.vocab, wbytes)
    else:
        mw = ModelWeights(cfg, spec, rank, nranks, me, embed, out_layers, final_norm, lm, rank * vloc, vloc, wbytes)
    mw.fp = fp
    mw.lm_split = lm_split
    return mw


class RcclComm:
    """RCCL communicator created through the C-ABI; the 128-byte unique id travels over the
    torch.distributed process group (plumbing)."""
    backend = "rccl"

    def __init__(self, rank, nranks, device):
        import torch.distributed as dist
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            check(lib().dihip_rccl_unique_id(buf), "dihip_rccl_unique_id")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        ident = ident.to(device)
        dist.broadcast(ident, 0)
        raw = bytes(ident.cpu().tolist())
        self.handle = C.c_void_p()
        check(lib().dihip_rccl_comm_init_rank(C.byref(self.handle), nranks, raw, rank), "dihip_rccl_comm_init_rank")
        self.rank, self.nranks = rank, nranks

    def allreduce_(self, t):
        check(lib().dihip_allreduce_sum(self.handle, ops.cur_stream(), ops.ptr(t), ops.ptr(t), t.numel(), ops.dt_code(t)),
              "dihip_allreduce_sum")
        return t

    def allgather(self, src, dst):
        check(lib().dihip_allgather_bytes(self.handle, ops.cur_stream(), ops.ptr(src), ops.ptr(dst),
                                          src.numel() * src.element_size()), "dihip_allgather_bytes")
        return dst


class TorchComm:
    """The same collectives through torch.distributed (backend nccl = RCCL): diagnostics only, selected explicitly with
    DIHIP_TP_ALLREDUCE=torch -- never a silent fallback."""
    backend = "torch.distributed"

    def __init__(self, rank, nranks):
        self.rank, self.nranks = rank, nranks

    def allreduce_(self, t):
        import torch.distributed as dist
        dist.all_reduce(t)
        return t

    def allgather(self, src, dst):
        import torch.distributed as dist
        dist.all_gather_into_tensor(dst, src)
        return dst


class P2PUnavailable(RuntimeError):
    """raised on EVERY rank (the ranks agree first) when the peer-to-peer all-reduce cannot be set up or fails its probe"""


def _all_ranks_ok(ok, device):
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


class P2PComm:
    """One-shot peer-to-peer all-reduce (csrc/p2p_allreduce.hip): every rank writes its row into every peer's receive buffer
    over xGMI; the receive buffers are exchanged as IPC handles over the torch.distributed process group (plumbing).
    The (rare, tiny) all-gather of the arg-max pairs and messages beyond the slot size go through `rccl`.

    guarded=True (make_comm backend "auto"): every local stage is followed by an agreement between the ranks, and the first
    exchanges run in the kernel's probe mode (a wait that gives up sets an error word instead of trapping), so that a node
    on which the IPC / peer path does not work ends in P2PUnavailable on all ranks -- not in a hang, a trap or a rank that
    sits alone in a collective."""
    backend = "p2p-oneshot"

    def __init__(self, rank, nranks, device, rccl, guarded=False):
        import torch.distributed as dist
        self.rank, self.nranks, self.rccl = rank, nranks, rccl
        l = lib()
        self.buf, self.handle, self.opened = C.c_void_p(), C.c_void_p(), []
        raw = (C.c_ubyte * 64)()

        def stage(what, fn):
            err = None
            try:
                fn()
            except Exception as e:  # noqa: BLE001
                if not guarded:
                    raise
                err = e
            if guarded and not _all_ranks_ok(err is None, device):
                raise P2PUnavailable(f"{what}: {err if err is not None else 'failed on another rank'}")

        def alloc():
            check(l.dihip_p2p_ar_alloc(C.byref(self.buf)), "dihip_p2p_ar_alloc")
            check(l.dihip_ipc_get_handle(self.buf, raw), "dihip_ipc_get_handle")
        stage("receive buffer / IPC handle", alloc)
        mine = torch.tensor(list(raw), dtype=torch.uint8, device=device)
        allh = [torch.zeros(64, dtype=torch.uint8, device=device) for _ in range(nranks)]
        dist.all_gather(allh, mine)

        def connect():
            ptrs = (C.c_void_p * nranks)()
            for r in range(nranks):
                if r == rank:
                    ptrs[r] = self.buf.value
                else:
                    p = C.c_void_p()
                    check(l.dihip_ipc_open_handle(bytes(allh[r].cpu().tolist()), C.byref(p)), "dihip_ipc_open_handle")
                    ptrs[r] = p.value
                    self.opened.append(p)
            check(l.dihip_p2p_ar_create(C.byref(self.handle), rank, nranks, ptrs), "dihip_p2p_ar_create")
        stage("opening the peers' buffers", connect)
        self.max_bytes = int(l.dihip_p2p_ar_max_bytes())
        dist.barrier()  # every rank has opened every buffer before the first push
        if guarded:
            def probe():
                check(l.dihip_p2p_ar_set_timeout(self.handle, 1 << 19, 0), "dihip_p2p_ar_set_timeout")  # ~ a second, no trap
                want = nranks * (nranks + 1) / 2
                for dt, n in ((torch.float32, 8), (torch.bfloat16, 3584)):
                    for _ in range(4):
                        t = torch.full((n,), float(rank + 1), dtype=dt, device=device)
                        self.allreduce_(t)
                        e = C.c_int(0)
                        check(l.dihip_p2p_ar_error(self.handle, C.byref(e)), "dihip_p2p_ar_error")
                        if e.value:
                            raise RuntimeError("a peer's row did not arrive (probe timeout)")
                        if abs(float(t[0]) - want) > 1e-3 or abs(float(t[-1]) - want) > 1e-3:
                            raise RuntimeError(f"wrong sum {float(t[0])} (expected {want})")
                check(l.dihip_p2p_ar_set_timeout(self.handle, 1 << 24, 1), "dihip_p2p_ar_set_timeout")
            stage("probe exchange", probe)

    verify = False   # set by bench.py around its eager verification steps: every peer-to-peer sum is checked against RCCL

    def allreduce_(self, t):
        nbytes = t.numel() * t.element_size()
        if nbytes > self.max_bytes or nbytes % 8:
            return self.rccl.allreduce_(t)
        ref = None
        if self.verify:
            ref = t.clone()
            self.rccl.allreduce_(ref)
        self._device_sum(t)
        if ref is not None:
            # (not under graph capture: it synchronises.)  Same addends, another summation order: a few ulps of the result type.
            r32, t32 = ref.float(), t.float()
            diff = float((t32 - r32).abs().max())
            scale = float(r32.abs().max())
            ulp = 2.0 ** -7 if t.dtype in (torch.bfloat16,) else (2.0 ** -10 if t.dtype == torch.float16 else 2.0 ** -20)
            self.verify_log.append((diff, scale, diff <= 4 * ulp * max(scale, 1e-6) * max(1, self.nranks // 2)))
        return t

    def _device_sum(self, t):
        check(lib().dihip_p2p_allreduce_sum(self.handle, ops.cur_stream(), ops.ptr(t), ops.ptr(t), t.numel(), ops.dt_code(t)),
              "dihip_p2p_allreduce_sum")

    def start_verification(self):
        self.verify, self.verify_log = True, []

    def finish_verification(self, device):
        """Ends a verification phase; every rank learns whether ALL ranks saw every peer-to-peer sum agree with RCCL's.
        Returns (ok, summary string)."""
        self.verify = False
        log = getattr(self, "verify_log", [])
        bad = [x for x in log if not x[2]]
        ok = _all_ranks_ok(len(bad) == 0 and len(log) > 0, device)
        worst = max((d / max(sc, 1e-6) for d, sc, _ in log), default=0.0)
        return ok, f"{len(log)} all-reduces checked against rccl, worst relative difference {worst:.2e}, {len(bad)} outside 4 ulp on this rank"

    def allgather(self, src, dst):
        return self.rccl.allgather(src, dst)


def make_comm(rank, nranks, device, backend=None, allow_labelled_fallback=False):
    """The tensor-parallel communicator, verified with one all-reduce.  backend (default: $DIHIP_TP_ALLREDUCE or "rccl"):
      "rccl"  ncclAllReduce over xGMI through the C-ABI (dihip_allreduce_sum)
      "p2p"   one-shot peer-to-peer all-reduce for decode-sized rows (dihip_p2p_allreduce_sum), RCCL for the rest
      "auto"  "p2p" if its guarded set-up and probe succeed on every rank (P2PComm(guarded=True)), else "rccl" -- and
              `.backend` says which and why (bench.py's default: the peer-to-peer path has never run between processes)
      "torch" torch.distributed collectives (diagnostics only)
    A backend that cannot be created or returns a wrong sum RAISES: a benchmark line must never come from a silently
    substituted path.  `.backend` names what runs.  allow_labelled_fallback (bench.py on a multi-GPU node it
    cannot be developed on): if the C-ABI communicator cannot be CREATED (e.g. the process already holds another RCCL
    build), torch.distributed's collectives are used and `.backend` says so in the JSON line -- substituted, never silently."""
    backend = backend or os.environ.get("DIHIP_TP_ALLREDUCE", "rccl")
    if backend == "torch":
        c = TorchComm(rank, nranks)
    elif backend in ("rccl", "p2p", "auto"):
        import sys
        try:
            c = RcclComm(rank, nranks, device)
        except Exception as e:  # noqa: BLE001
            if not allow_labelled_fallback:
                raise
            print(f"[rank {rank}] C-ABI RCCL communicator could not be created: {e}; torch.distributed collectives instead "
                  "(recorded in comm_backend)", file=sys.stderr)
            c = TorchComm(rank, nranks)
            c.backend = f"torch.distributed (labelled fallback: rccl communicator failed: {type(e).__name__})"
        base = c  # also serves the peer-to-peer communicator's all-gather and its over-long messages
        if backend == "p2p":
            c = P2PComm(rank, nranks, device, base)
        elif backend == "auto":
            try:
                c = P2PComm(rank, nranks, device, base, guarded=True)
                if base.backend != "rccl":
                    c.backend = f"p2p-oneshot (all-gather / long messages: {base.backend})"
            except P2PUnavailable as e:
                print(f"[rank {rank}] peer-to-peer all-reduce unavailable ({e}); {base.backend} instead (recorded in comm_backend)",
                      file=sys.stderr)
                c.backend = f"{base.backend} (p2p-oneshot unavailable: {str(e)[:120]})"
    else:
        raise ValueError(f"unknown tensor-parallel all-reduce backend {backend!r}")
    for dt, n in ((torch.float32, 8), (torch.bfloat16, 3584)):
        probe = torch.full((n,), float(rank + 1), dtype=dt, device=device)
        for _ in range(3):  # repeated: the one-shot protocol alternates two slot sets
            probe.fill_(float(rank + 1))
            c.allreduce_(probe)
            if torch.device(device).type == "cuda":
                torch.cuda.synchronize()
            if abs(float(probe[0]) - nranks * (nranks + 1) / 2) > 1e-3 or abs(float(probe[-1]) - nranks * (nranks + 1) / 2) > 1e-3:
                raise RuntimeError(f"[rank {rank}] all-reduce backend {c.backend} returned a wrong sum ({float(probe[0])})")
    return c


class DecodeSession:
    """Buffers + KV spans for a fixed batch; `step()` enqueues one decode step."""

    def __init__(self, model: ModelWeights, batch, max_len, span_len=128, kv_mode="none", comm: Optional[RcclComm] = None,
                 device="cuda", ar_overlap: Optional[bool] = None):
        cfg = model.cfg
        self.model, self.B, self.max_len, self.comm = model, batch, max_len, comm
        self.kv_mode = kv_mode
        H = cfg.head_dim
        self.n_loc, self.g_loc, self.H = len(model.heads.q_heads), len(model.heads.kv_heads), H
        dt = model.embed.dtype
        L = len(model.layers)
        spans_per_req = (max_len + span_len - 1) // span_len
        self.pool = ops.SpanPool(2 * L * batch * spans_per_req + 1, self.g_loc, span_len, H, kv_mode, dt, device)
        self.kv = [ops.KVCacheSet(self.pool, batch, spans_per_req) for _ in range(L)]
        for kv in self.kv:  # all spans are claimed up front: the span tables are static under graph replay
            for b in range(batch):
                kv.ensure(b, max_len)
            kv.sync()
        self.inv_freq = (1.0 / (cfg.rope_theta ** (torch.arange(0, H, 2, dtype=torch.float64) / H))).float().to(device)
        f32 = torch.float32
        self.ids = torch.zeros(batch, dtype=torch.int64, device=device)
        self.old_lens = torch.zeros(batch, dtype=torch.int32, device=device)
        self.new_lens = torch.ones(batch, dtype=torch.int32, device=device)
        self.h = torch.empty(batch, cfg.hidden, dtype=f32, device=device)
        self.qkv = torch.empty(batch, (self.n_loc + 2 * self.g_loc) * H, dtype=dt, device=device)
        self.q = torch.empty(batch, self.n_loc * H, dtype=dt, device=device)
        self.attn = torch.empty(batch, self.n_loc * H, dtype=dt, device=device)
        if cfg.moe is not None:
            mc, i_loc = cfg.moe, model.layers[0].gate.N
            self.moe_xn = torch.empty(batch, cfg.hidden, dtype=dt, device=device)
            self.moe_logits = torch.empty(batch, mc.num_experts, dtype=dt, device=device)
            self.moe_scores = torch.empty(batch, mc.top_k, dtype=f32, device=device)
            self.moe_experts = torch.empty(batch, mc.top_k, dtype=torch.int32, device=device)
            self.moe_out = torch.empty(batch, cfg.hidden, dtype=dt, device=device)
            self.moe_act = torch.empty(batch, i_loc, dtype=dt, device=device)
            # small batches hand SiLU(gate) * up to the down projection in the MFMA-fragment layout: the K-slice / panel kernels
            # read every activation once per workgroup K-range, the whole-column kernel re-reads all of it per column tile
            # (224 x 655 KB at 16 x 20480: 28.3 us for 73 MB, profiles/r03w)
            self.moe_act_frag = (torch.zeros(ops.act_frag_numel(batch, i_loc), dtype=dt, device=device)
                                 if 4 < batch <= 32 and i_loc % 32 == 0 and os.environ.get("DIHIP_MOE_ACT_FRAG", "1") != "0" else None)
            self.moe_shared = torch.empty(batch, cfg.hidden, dtype=dt, device=device)
            self.moe_sig = torch.empty(batch, 1, dtype=dt, device=device)
            self.moe_ws = torch.empty(int(lib().dihip_moe_workspace_bytes(batch, mc.top_k, cfg.hidden, mc.moe_inter)), dtype=torch.uint8, device=device)
            self.moe_dense_scratch = ops.Scratch(int(lib().dihip_dense_workspace_bytes(batch, mc.num_experts, cfg.hidden)), device)
        self.attn_frag = False  # set below: the op-boundary attention can write the o-projection's fragment layout
        # the SwiGLU output only feeds the down projection: when both run on the small-batch kernel it is
        # kept in the MFMA-fragment layout that kernel loads with contiguous 1 KiB wave-loads
        l0_ = model.layers[0]
        self.act_frag = ops.prefers_frag(l0_.gate, batch, dual=True) and ops.prefers_frag(l0_.down, batch)
        if self.act_frag:
            self.act = torch.zeros(ops.act_frag_numel(batch, l0_.gate.N), dtype=dt, device=device)
        else:
            self.act = torch.empty(batch, l0_.gate.N, dtype=dt, device=device)
        self.logits = torch.empty(batch, model.vocab_local, dtype=f32, device=device)
        self.lm_ksplit = getattr(model, "lm_split", "vocab") == "k" and model.nranks > 1
        if self.lm_ksplit:
            self.lm_xn = torch.empty(batch, cfg.hidden, dtype=dt, device=device)
            kloc = cfg.hidden // model.nranks
            self.lm_xs = torch.empty(batch, kloc, dtype=dt, device=device)
        self.partial = torch.zeros(batch, cfg.hidden, dtype=f32, device=device)
        l0, wb, gsz = model.layers[0], model.quant.wbits, model.quant.group
        need = max(ops.lowp_workspace_bytes(wb, batch, p.N, p.K, gsz) for p in (l0.qkv, l0.o, l0.gate, l0.down))
        need = max(need, int(lib().dihip_dense_workspace_bytes(batch, model.lm_head.N, model.lm_head.K)))
        self.scratch = ops.Scratch(need, device)
        self.attn_ws = torch.empty(max(ops.span_attn_workspace(batch, self.n_loc, H, max_len),
                                       ops.span_attn_fused_workspace(batch, self.n_loc, self.g_loc, H, max_len), 256),
                                   dtype=torch.uint8, device=device)
        self.attn_sync = torch.zeros(int(lib().dihip_span_attn_sync_bytes(batch, self.n_loc)), dtype=torch.uint8, device=device)
        self.rope_tab = ops.rope_table(self.inv_freq, max_len + 1, H)
        # 16-bit cache: Rotary + cache append + attention in ONE launch (+ the split merge), both contractions on the matrix
        # cores, at every batch size.  Quantised caches: the quantising append launch, then the matrix-core decode kernels
        # (dihip_span_attn_decode_fused would issue the same two launches; the explicit pair also has the FRAG32 output).
        self.fused_attention = kv_mode == "none"
        # uint4 cache with bf16 rows (round 4): one launch as well, FRAG32 output included (dihip_span_attn_decode_step);
        # DIHIP_ATTN_U4_FUSED=0 keeps the append launch (A/B)
        self.step_attention = ((kv_mode == "u4" and dt == torch.bfloat16 and os.environ.get("DIHIP_ATTN_U4_FUSED", "1") != "0")
                               # int8 cache: one launch too, where it pays (batch 1: -1.8 us per layer; batch 32: +1.3 -- every workgroup
                               # rotates its query heads itself; profiles/r05_i8_decode_step.txt)
                               or (kv_mode == "i8" and batch <= 4 and os.environ.get("DIHIP_ATTN_I8_FUSED", "1") != "0"))
        # split sequences: the partial records are merged inside the attention launch (arrival tickets in attn_sync, zeroed
        # once) instead of by a second launch; DIHIP_DECODER_ATTN_MERGE=launch restores the two-launch form (A/B)
        self.attn_merge_in_launch = os.environ.get("DIHIP_DECODER_ATTN_MERGE", "ticket") != "launch"
        # Batch 1 (round 5): RMSNorm + qkv GEMV, Rotary + append + attention + merge and the o-projection + residual as ONE launch
        # (dihip_decode_attn_block: weights and K / V tiles requested at launch, the operators hand over in-launch); bit-identical
        # to the three launches it replaces.  DIHIP_DECODER_ATTN_BLOCK=0 keeps the chain (A/B).
        self.attn_block = (batch == 1 and self.fused_attention and self.attn_merge_in_launch
                           and os.environ.get("DIHIP_DECODER_ATTN_BLOCK", "1") != "0"
                           and ops.decode_attn_block_supported(model.layers[0].qkv, cfg.hidden, self.n_loc, self.g_loc, H, max_len, kv_mode, dt, batch))
        if self.attn_block:
            need_ws = int(lib().dihip_decode_attn_block_workspace_bytes(self.n_loc, self.g_loc, H, max_len))
            if self.attn_ws.numel() < need_ws:
                self.attn_ws = torch.empty(need_ws, dtype=torch.uint8, device=device)
            self.block_sync = torch.zeros(int(lib().dihip_decode_attn_block_sync_bytes(self.n_loc, self.g_loc, H)), dtype=torch.uint8, device=device)
        # The feed-forward half -- RMSNorm + gate / up + SwiGLU and the down projection + residual -- as ONE launch as well
        # (dihip_decode_mlp_block: two launches per layer) is bit-identical but NOT faster: 26.6 vs 25.8 us per layer, because the
        # decode GEMV consumes a resident chunk no faster than HBM delivers one, so the 64 KB per workgroup prefetched across the
        # hand-off buy nothing and the hand-off (3.7 us) costs more than the boundary it replaces (profiles/r05_mlp_block_timeline.txt).
        # OFF by default; DIHIP_DECODER_MLP_BLOCK=1 runs it (A/B, tests).
        self.mlp_block = (batch == 1 and cfg.moe is None and os.environ.get("DIHIP_DECODER_MLP_BLOCK", "0") == "1"
                          and ops.decode_mlp_block_supported(model.layers[0].gate, cfg.hidden, dt, batch))
        if self.mlp_block:
            self.mlp_sync = torch.zeros(int(lib().dihip_decode_mlp_block_sync_bytes(model.layers[0].gate.N)), dtype=torch.uint8, device=device)
        if not self.fused_attention and batch <= 32 and ops.prefers_frag(model.layers[0].o, batch):
            self.attn_frag = True
            self.attn = torch.zeros(ops.act_frag_numel(batch, self.n_loc * H), dtype=dt, device=device)
        # Batched decode (4 < batch <= 32, one rank): the RMSNorm after each residual GEMM is produced by that GEMM call
        # (riding on its split-K reduction) and handed to the next GEMM in its preferred layout -- two launches less
        # per layer.  Under TP the all-reduce sits between the GEMM and the norm, so the separate norm stays.
        self.norm_fuse = (4 < batch <= 32 and (comm is None or model.nranks == 1) and cfg.moe is None
                          and os.environ.get("DIHIP_DECODER_NORM_FUSE", "1") != "0")  # =0: diagnostics
        if self.norm_fuse:
            def norm_buf(pw, dual):
                frag = ops.prefers_frag(pw, batch, dual=dual)
                n = ops.act_frag_numel(batch, pw.K) if frag else batch * pw.K
                return torch.zeros(n, dtype=dt, device=device), (ops.ACT_FRAG32 if frag else ops.ACT_ROWMAJOR)
            self.xn1, self.xn1_layout = norm_buf(l0_.qkv, False)
            self.xn2, self.xn2_layout = norm_buf(l0_.gate, True)
        # ... and where the consumer's kernel takes it, the norm is not even computed as such: the producer leaves FT(gamma * h) and
        # partial sums of h^2, the consumer scales its accumulators by 1 / rms (ops.fused_gemm_addto_prenorm; bf16; DIHIP_DEFER_RMSNORM=0 off).
        # 16-bit cache only: the moved rounding point is an independent rounding of the reference's size (same error against the
        # mathematics, tests/test_gpu_deferred_norm.py), and the quantised caches' tests sit at their asserted logit bound already.
        defer_ok = self.norm_fuse and dt == torch.bfloat16 and kv_mode == "none"
        self.defer_ln2 = bool(defer_ok and ops.prenorm_rowsq_supported(l0_.gate, batch, dual=True, x_layout=self.xn2_layout))
        self.defer_ln1 = bool(defer_ok and ops.prenorm_rowsq_supported(l0_.qkv, batch, dual=False, x_layout=self.xn1_layout))
        self.rowsq1 = ops.rowsq_buffer(device) if self.defer_ln1 else None
        self.rowsq2 = ops.rowsq_buffer(device) if self.defer_ln2 else None
        self.rowsq1_parts = 0
        # Tensor-parallel all-reduce schedule.  Default: on the compute stream.  DIHIP_TP_OVERLAP=1: the north star's schedule
        # -- the collective on a side HIP stream between two events (record after the producing GEMV -> side stream waits ->
        # all-reduce -> record -> compute stream waits), while the compute stream pulls the weights of the NEXT GEMV into
        # the Infinity Cache (the only work of the decode chain that does not depend on the reduced row).
        want_overlap = os.environ.get("DIHIP_TP_OVERLAP", "0") == "1" if ar_overlap is None else bool(ar_overlap)
        self.ar_overlap = comm is not None and model.nranks > 1 and want_overlap
        self.side_stream = torch.cuda.Stream(device=device) if self.ar_overlap else None
        self.argmax_ws = torch.empty(batch * 64 * 8, dtype=torch.uint8, device=device)
        nr = model.nranks
        self.pair = torch.empty(batch * 8, dtype=torch.uint8, device=device)
        self.pairs_all = torch.empty(nr * batch * 8, dtype=torch.uint8, device=device)
        self.scale = 1.0 / (H ** 0.5)
        self.graph = None

    # -- state ---------------------------------------------------------------------------
    def set_state(self, ids, lens):
        """ids: next input token per request; lens: tokens already in the cache."""
        self.ids.copy_(torch.as_tensor(ids, dtype=torch.int64))
        lens = torch.as_tensor(lens, dtype=torch.int32)
        self.old_lens.copy_(lens)
        self.new_lens.copy_(lens + 1)

    def fill_cache_random(self, length, seed=7):
        """Synthetic KV history of `length` tokens (K,V ~ N(0,1))."""
        gen = torch.Generator(device=self.pool.pool.device)
        gen.manual_seed(seed)
        if self.kv_mode == "none":
            v = self.pool.pool.view(self.model.embed.dtype)
            v.copy_(torch.randn(v.shape, generator=gen, device=v.device, dtype=torch.float32).to(v.dtype))
        else:
            self.pool.pool.random_(0, 256, generator=gen)
            hb = self.H if self.kv_mode == "i8" else self.H // 2
            g, S = self.g_loc, self.pool.S
            per = self.pool.aligned
            allp = self.pool.pool.view(-1, per)[:, g * S * hb: g * S * hb + g * S * 8].contiguous().view(torch.float32).view(-1, g, S, 2)
            allp[..., 0] = 8.0 if self.kv_mode == "u4" else 0.0
            allp[..., 1] = 0.25 if self.kv_mode == "u4" else 0.02
            self.pool.pool.view(-1, per)[:, g * S * hb: g * S * hb + g * S * 8] = allp.view(torch.uint8).view(-1, g * S * 8)

    # -- context (prefill) phase ----------------------------------------------------------------
    def prefill(self, seqs, return_logits=True):
        """Context phase of every request through the product's op-boundary entry points, in the order of the reference
        graph (qwen_v15.py:210-388; SpanAttnOp::runContext, span_attn_op_cuda.cpp:205-330): embedding -> per layer
        RMSNorm + qkv GEMM(+bias) -> Rotary on the fused rows -> ContextSpanCopy of K, V into the request's spans ->
        causal prefill attention -> o GEMM + residual -> RMSNorm + gate/up GEMM + SwiGLU -> down GEMM + residual;
        final norm + lm_head on the last row.  Leaves the session ready to decode: lens = len(seq), ids = greedy next
        token.  seqs: one token-id list per request (single rank)."""
        m, cfg = self.model, self.model.cfg
        assert m.nranks == 1 and len(seqs) == self.B
        n, g, H, dev = self.n_loc, self.g_loc, self.H, self.h.device
        Lmax = max(len(s) for s in seqs)
        l0, wb, gsz = m.layers[0], m.quant.wbits, m.quant.group
        # per prompt length: a shorter prompt can take a split-K plan with a larger slab than the longest one
        need = max(ops.lowp_workspace_bytes(wb, L, p.N, p.K, gsz) for L in {len(s) for s in seqs} for p in (l0.qkv, l0.o, l0.gate, l0.down))
        need = max(need, int(lib().dihip_dense_workspace_bytes(1, m.lm_head.N, m.lm_head.K)))
        sc = ops.Scratch(need, dev)
        logits = torch.empty(self.B, m.vocab_local, dtype=torch.float32, device=dev) if return_logits else None
        for b, seq in enumerate(seqs):
            L = len(seq)
            assert 0 < L <= self.max_len
            ids = torch.as_tensor(seq, dtype=torch.int64, device=dev)
            pos = torch.arange(L, dtype=torch.int32, device=dev)
            h = ops.embedding(ids, m.embed)
            for li, lw in enumerate(m.layers):
                qkv = ops.fused_norm_gemm(h, lw.ln1, cfg.eps, lw.qkv, lw.qkv_bias, sc)
                ops.rope_qk_(qkv, pos, self.inv_freq, n, g, H)
                kv = self.kv[li]
                stride = (n + 2 * g) * H
                ops.kv_context_copy(kv.k_ptrs[b], qkv[:, n * H:], stride, L, 0, g, H, self.pool.S, self.kv_mode)
                ops.kv_context_copy(kv.v_ptrs[b], qkv[:, (n + g) * H:], stride, L, 0, g, H, self.pool.S, self.kv_mode)
                attn = ops.prefill_attn(qkv[:, : n * H], qkv[:, n * H:(n + g) * H], qkv[:, (n + g) * H:], n, g, H, self.scale)
                h = ops.fused_gemm_addto(attn, lw.o, h, sc, M=L)
                if cfg.moe is not None:
                    h = self._moe_rows(h, lw, sc)
                    continue
                act = ops.fused_norm_swiglu(h, lw.ln2, cfg.eps, lw.gate, lw.up, sc)
                h = ops.fused_gemm_addto(act, lw.down, h, sc, M=L)
            if return_logits:
                ops.lm_head(h[L - 1:L].contiguous(), m.final_norm, cfg.eps, m.lm_head, sc, out=logits[b:b + 1])
        lens = [len(s) for s in seqs]
        if return_logits:
            nxt = torch.argmax(logits, dim=-1).cpu()
            self.set_state(nxt, lens)
        else:
            self.set_state(torch.zeros(self.B, dtype=torch.int64), lens)
        return logits

    def _moe_rows(self, h, lw, sc):
        """The mixture-of-experts block (_moe_block below) over the T rows of a prompt, on fresh tensors, single rank."""
        cfg, mc = self.model.cfg, self.model.cfg.moe
        T, proj = h.shape[0], lw.exp_gate.N
        xn = ops.rmsnorm_rows(h, lw.ln2, cfg.eps)
        logits, sig = ops.moe_router_gate(xn, lw.router, lw.shared_sig)
        ws = torch.empty(int(lib().dihip_moe_workspace_bytes(T, mc.top_k, cfg.hidden, mc.moe_inter)), dtype=torch.uint8, device=h.device)
        grouped = T > 1 and T * mc.top_k <= ops.MOE_GROUP_MAX_SLOTS and os.environ.get("DIHIP_MOE_FUSED", "1") != "0"
        if grouped:
            scores, experts = ops.moe_route_grouped(logits, mc.top_k, cfg.hidden, proj, ws, ep=lw.ep)
        else:
            scores, experts = ops.moe_route(logits, mc.top_k, ep=lw.ep)
        out = ops.moe_experts(xn, experts, scores, lw.exp_gate, lw.exp_up, lw.exp_down, ws=ws,
                              out=torch.empty(T, cfg.hidden, dtype=xn.dtype, device=h.device),
                              flags=(ops.MOE_PREGROUPED | ops.MOE_NO_FINALIZE) if grouped else 0)
        act = ops.prenorm_swiglu(xn, lw.gate, lw.up, sc, T)
        shared = ops.gemm_lowp(act, lw.down, scratch=sc)
        if grouped:
            return ops.moe_combine(h, h, ws, scores, experts, shared, sig, proj)
        return ops.moe_shared_combine(h, h, out, shared, sig)

    # -- one decode step -------------------------------------------------------------------
    def step(self):
        m = self.model
        ops.embedding(self.ids, m.embed, out=self.h)
        L = len(m.layers)
        for li in range(L):
            self._layer(li, first=li == 0, last=li + 1 == L)
        self._head()

    def run_single_layer(self, li):
        """Layer li alone on the f32 hidden rows in self.h (in place), against the cache state (old_lens) as it is: the
        per-layer teacher-forced drift test feeds the oracle's layer input here (tests/test_gpu_parity_depth.py).  Same
        launches as the layer inside step(), in the forms the first layer (norm from h) and the last layer (no norm handed
        on) take -- bit-identical to the mid-layer forms (tests/test_gpu_gemm.py)."""
        self._layer(li, first=True, last=True)

    def _layer(self, li, first, last):
        m, cfg, sc = self.model, self.model.cfg, self.scratch
        lw = m.layers[li]
        tp_on = self.comm is not None and m.nranks > 1
        nf = self.norm_fuse and not tp_on
        if self.attn_block:
            h_res = self.h if (not tp_on or m.rank == 0) else None
            ops.decode_attn_block(self.h, h_res, lw.ln1, cfg.eps, lw.qkv, lw.qkv_bias, lw.o, self.kv[li], self.old_lens, self.rope_tab,
                                  self.n_loc, self.g_loc, self.H, self.max_len, self.scale, self.attn_ws, self.block_sync, out=self.h)
            if tp_on:
                self._allreduce(self.h, () if cfg.moe is not None else (lw.gate.w, lw.up.w))
            if cfg.moe is not None:
                self._moe_block(lw, tp_on)
                return
            self._mlp(li, tp_on)
            return
        if nf and not first and self.rowsq1_parts:
            ops.prenorm_gemm_rowsq(self.xn1, lw.qkv, lw.qkv_bias, sc, self.B, self.rowsq1, self.rowsq1_parts, cfg.eps, x_layout=self.xn1_layout,
                                   out=self.qkv)
        elif nf and not first:
            ops.prenorm_gemm(self.xn1, lw.qkv, lw.qkv_bias, sc, self.B, x_layout=self.xn1_layout, out=self.qkv)
        else:
            ops.fused_norm_gemm(self.h, lw.ln1, cfg.eps, lw.qkv, lw.qkv_bias, sc, out=self.qkv)
        if self.fused_attention:
            ops.span_attn_decode_fused(self.qkv, self.kv[li], self.old_lens, self.rope_tab, self.n_loc, self.g_loc, self.H,
                                       self.max_len, self.scale, self.attn_ws, out=self.attn,
                                       sync=self.attn_sync if self.attn_merge_in_launch else None)
        elif self.step_attention:
            ops.span_attn_decode_step(self.qkv, self.kv[li], self.old_lens, self.rope_tab, self.n_loc, self.g_loc, self.H, self.max_len,
                                      self.scale, self.attn_ws, self.attn_sync if self.attn_merge_in_launch else None, out=self.attn,
                                      out_layout=ops.ACT_FRAG32 if self.attn_frag else ops.ACT_ROWMAJOR)
        else:
            ops.rope_kv_append(self.kv[li], self.q, self.qkv, self.old_lens, self.inv_freq, self.n_loc, self.g_loc, self.H)
            ops.span_attn_decode(self.q, self.kv[li], self.new_lens, self.n_loc, self.g_loc, self.H, self.max_len,
                                 self.scale, self.attn_ws, self.attn_sync, out=self.attn,
                                 out_layout=ops.ACT_FRAG32 if self.attn_frag else ops.ACT_ROWMAJOR)
        if cfg.moe is not None:
            self._proj_residual(self.attn, lw.o, tp_on, frag=self.attn_frag)
            self._moe_block(lw, tp_on)
            return
        if nf:
            parts2 = 0
            if self.defer_ln2:
                _, parts2 = ops.fused_gemm_addto_prenorm(self.attn, lw.o, self.h, sc, lw.ln2, cfg.eps, self.xn2, self.rowsq2, out=self.h,
                                                         x_layout=ops.ACT_FRAG32 if self.attn_frag else ops.ACT_ROWMAJOR,
                                                         xnorm_layout=self.xn2_layout, M=self.B)
            else:
                ops.fused_gemm_addto_norm(self.attn, lw.o, self.h, sc, lw.ln2, cfg.eps, self.xn2, out=self.h,
                                          x_layout=ops.ACT_FRAG32 if self.attn_frag else ops.ACT_ROWMAJOR,
                                          xnorm_layout=self.xn2_layout, M=self.B)
            if parts2:
                ops.prenorm_swiglu_rowsq(self.xn2, lw.gate, lw.up, sc, self.B, self.rowsq2, parts2, cfg.eps, x_layout=self.xn2_layout, out=self.act,
                                         y_layout=ops.ACT_FRAG32 if self.act_frag else ops.ACT_ROWMAJOR)
            else:
                ops.prenorm_swiglu(self.xn2, lw.gate, lw.up, sc, self.B, x_layout=self.xn2_layout, out=self.act,
                                   y_layout=ops.ACT_FRAG32 if self.act_frag else ops.ACT_ROWMAJOR)
            self.rowsq1_parts = 0
            if not last and self.defer_ln1:
                _, self.rowsq1_parts = ops.fused_gemm_addto_prenorm(self.act, lw.down, self.h, sc, m.layers[li + 1].ln1, cfg.eps, self.xn1,
                                                                    self.rowsq1, out=self.h,
                                                                    x_layout=ops.ACT_FRAG32 if self.act_frag else ops.ACT_ROWMAJOR,
                                                                    xnorm_layout=self.xn1_layout, M=self.B)
            elif not last:
                ops.fused_gemm_addto_norm(self.act, lw.down, self.h, sc, m.layers[li + 1].ln1, cfg.eps, self.xn1, out=self.h,
                                          x_layout=ops.ACT_FRAG32 if self.act_frag else ops.ACT_ROWMAJOR,
                                          xnorm_layout=self.xn1_layout, M=self.B)
            else:
                self._proj_residual(self.act, lw.down, tp_on, frag=self.act_frag)  # lm_head applies the final norm itself
            return
        self._proj_residual(self.attn, lw.o, tp_on, frag=self.attn_frag, next_weights=(lw.gate.w, lw.up.w))
        self._mlp(li, tp_on)

    def _mlp(self, li, tp_on):
        """the dense feed-forward half of layer li on self.h (in place), + its all-reduce under TP"""
        m, cfg, sc = self.model, self.model.cfg, self.scratch
        lw = m.layers[li]
        nxt = m.layers[li + 1].qkv if li + 1 < len(m.layers) else m.lm_head
        if self.mlp_block:
            h_res = self.h if (not tp_on or m.rank == 0) else None
            ops.decode_mlp_block(self.h, h_res, lw.ln2, cfg.eps, lw.gate, lw.up, lw.down, self.mlp_sync, out=self.h)
            if tp_on:
                self._allreduce(self.h, (nxt.w,))
            return
        ops.fused_norm_swiglu(self.h, lw.ln2, cfg.eps, lw.gate, lw.up, sc, out=self.act,
                              y_layout=ops.ACT_FRAG32 if self.act_frag else ops.ACT_ROWMAJOR)
        self._proj_residual(self.act, lw.down, tp_on, frag=self.act_frag, next_weights=(nxt.w,))

    def _head(self):
        m, cfg, sc = self.model, self.model.cfg, self.scratch
        tp_on = self.comm is not None and m.nranks > 1
        if self.lm_ksplit and tp_on:
            # the reference's lm_head under TP (model_base.py:690-703): final norm -> this rank's K slice of the row times its row
            # block of the weight -> all-reduce of the partial logits: every rank ends with the full f32 logits row
            ops.rmsnorm_rows(self.h, m.final_norm, cfg.eps, out=self.lm_xn)
            kloc = self.lm_xs.shape[1]
            self.lm_xs.copy_(self.lm_xn[:, m.rank * kloc:(m.rank + 1) * kloc])
            ops.fused_gemm_addto(self.lm_xs, m.lm_head, None, sc, out=self.logits, M=self.B)
            self._allreduce(self.logits)
            ops.argmax(self.logits, ws=self.argmax_ws, out=self.ids, advance=(self.old_lens, self.new_lens))
            return
        ops.lm_head(self.h, m.final_norm, cfg.eps, m.lm_head, sc, out=self.logits)
        if tp_on:
            check(lib().dihip_argmax_partial(ops.cur_stream(), ops.ptr(self.pair), ops.ptr(self.logits), self.B,
                                             m.vocab_local, m.vocab_offset, ops.ptr(self.argmax_ws), self.argmax_ws.numel()),
                  "dihip_argmax_partial")
            self.comm.allgather(self.pair, self.pairs_all)
            check(lib().dihip_argmax_merge(ops.cur_stream(), ops.ptr(self.ids), ops.ptr(self.pairs_all), m.nranks, self.B),
                  "dihip_argmax_merge")
            ops.increment_u32_(self.old_lens)
            ops.increment_u32_(self.new_lens)
        else:
            ops.argmax(self.logits, ws=self.argmax_ws, out=self.ids, advance=(self.old_lens, self.new_lens))

    def _moe_block(self, lw, tp_on):
        """The mixture-of-experts feed-forward block (python/pyhie/allspark/model/qwen_v20_moe.py:318-382):
          ffn_ln -> mlp.gate (router Gemm) -> MOE (softmax, top-k, this rank's experts, combine)
                 -> shared_expert gate_up Gemm + UnaryGLU -> down Gemm ; shared_expert_gate (Gemm, SIGMOID) ; CalcExpert
                 -> expert_add, final_add  [-> all-reduce]
        Under expert parallelism (attribute use_ep) a rank runs the experts it owns and the shared expert's column / row
        slices; the reference all-reduces the MOE output and the CalcExpert output separately, here the two partial sums and
        the residual (rank 0) are added first and the f32 hidden rows are all-reduced once -- the same sum."""
        cfg, m, sc, B = self.model.cfg, self.model, self.scratch, self.B
        ops.rmsnorm_rows(self.h, lw.ln2, cfg.eps, out=self.moe_xn)
        if fused_dense := os.environ.get("DIHIP_MOE_FUSED", "1") != "0":   # router + shared-expert gate Gemms in one launch
            ops.moe_router_gate(self.moe_xn, lw.router, lw.shared_sig, logits=self.moe_logits, sig=self.moe_sig)
        else:
            ops.gemm_dense(self.moe_xn, lw.router, out=self.moe_logits, scratch=self.moe_dense_scratch)
        # decode batches: routing + slot grouping in one launch, finalize-routing folded into the combine (with the fused dense
        # pair: 8 launches per block instead of 11; DIHIP_MOE_FUSED=0: the separate calls)
        fused = B > 1 and B * cfg.moe.top_k <= ops.MOE_GROUP_MAX_SLOTS and os.environ.get("DIHIP_MOE_FUSED", "1") != "0"
        if fused:
            ops.moe_route_grouped(self.moe_logits, cfg.moe.top_k, cfg.hidden, lw.exp_gate.N, self.moe_ws, ep=lw.ep,
                                  scores=self.moe_scores, experts=self.moe_experts)
        else:
            ops.moe_route(self.moe_logits, cfg.moe.top_k, ep=lw.ep, scores=self.moe_scores, experts=self.moe_experts)
        if getattr(self, "_expert_log", None) is not None:
            self._expert_log.append(self.moe_experts.clone())
        ops.moe_experts(self.moe_xn, self.moe_experts, self.moe_scores, lw.exp_gate, lw.exp_up, lw.exp_down, ws=self.moe_ws, out=self.moe_out,
                        flags=(ops.MOE_PREGROUPED | ops.MOE_NO_FINALIZE) if fused else 0)
        if self.moe_act_frag is not None:
            ops.prenorm_swiglu(self.moe_xn, lw.gate, lw.up, sc, B, out=self.moe_act_frag, y_layout=ops.ACT_FRAG32)
            ops.prenorm_gemm(self.moe_act_frag, lw.down, None, sc, B, x_layout=ops.ACT_FRAG32, out=self.moe_shared)
        else:
            ops.prenorm_swiglu(self.moe_xn, lw.gate, lw.up, sc, B, out=self.moe_act)
            ops.gemm_lowp(self.moe_act, lw.down, scratch=sc, out=self.moe_shared)
        if not fused_dense:
            ops.gemm_dense(self.moe_xn, lw.shared_sig, act="sigmoid", out=self.moe_sig, scratch=self.moe_dense_scratch)
        h_res = self.h if (not tp_on or m.rank == 0) else None
        if fused:
            ops.moe_combine(self.h, h_res, self.moe_ws, self.moe_scores, self.moe_experts, self.moe_shared, self.moe_sig, lw.exp_gate.N)
        else:
            ops.moe_shared_combine(self.h, h_res, self.moe_out, self.moe_shared, self.moe_sig)
        if tp_on:
            self._allreduce(self.h)

    def _allreduce(self, t, next_weights=()):
        """Sum all-reduce of the hidden rows; with the overlap schedule on the side stream, beside a cache prefetch of the
        weights the next launch streams."""
        if not self.ar_overlap:
            self.comm.allreduce_(t)
            return
        cur = torch.cuda.current_stream()
        produced = torch.cuda.Event()
        produced.record(cur)
        self.side_stream.wait_event(produced)
        with torch.cuda.stream(self.side_stream):
            self.comm.allreduce_(t)
            reduced = torch.cuda.Event()
            reduced.record(self.side_stream)
        if next_weights:
            ops.prefetch([w for w in next_weights if w is not None], workgroups=64)
        cur.wait_event(reduced)

    def _proj_residual(self, x, pw, tp_on, frag=False, next_weights=()):
        """h += x . W  (row-parallel under TP: rank 0 carries the residual, then all-reduce --
        the reference applies the fused residual ADD on rank 0 only, gemm_op.cpp:133-137)."""
        lay = ops.ACT_FRAG32 if frag else ops.ACT_ROWMAJOR
        h_res = self.h if (not tp_on or self.model.rank == 0) else None
        ops.fused_gemm_addto(x, pw, h_res, self.scratch, out=self.h, x_layout=lay, M=self.B)
        if tp_on:
            self._allreduce(self.h, next_weights)

    # -- hipGraph capture --------------------------------------------------------------------
    def capture(self, warmup=2, steps_per_graph=1):
        """Capture one decode step as a hipGraph (self.graph) and, for steps_per_graph > 1, a second graph of that
        many consecutive steps (self.graph_multi): greedy decoding needs no host decision between steps (ids and
        lengths live on the device), so k steps replay with one graph launch instead of k."""
        ids0, old0, new0 = self.ids.clone(), self.old_lens.clone(), self.new_lens.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.ids.copy_(ids0)
        self.old_lens.copy_(old0)
        self.new_lens.copy_(new0)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.step()
        self.graph_multi, self.steps_per_graph = None, 1
        if steps_per_graph > 1:
            self.graph_multi = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_multi):
                for _ in range(steps_per_graph):
                    self.step()
            self.steps_per_graph = steps_per_graph
        torch.cuda.synchronize()
        # the capture pass itself does not execute; state is still (ids0, lens0)
        return self.graph

    def check_handoffs(self):
        """At a synchronisation point of the caller: the fused attention block's error word (word 1 of its sync buffer, set when a bounded
        hand-off wait of a launch gave up).  Non-zero -> the steps since the last check are invalid: the buffer's invariants are restored,
        the session keeps the three-launch chain from here (captured graphs are dropped: capture again) and RuntimeError is raised --
        never silently wrong tokens.  The C++ runner does the same in HipModelRunner::Sync."""
        if not getattr(self, "attn_block", False):
            return
        torch.cuda.synchronize()
        code = int(self.block_sync.view(torch.int32)[1].item())
        if code:
            ops.check(lib().dihip_decode_attn_block_reset(ops.cur_stream(), ops.ptr(self.block_sync), self.block_sync.numel()), "attn_block_reset")
            torch.cuda.synchronize()
            self.attn_block = False
            self.graph = self.graph_multi = None
            raise RuntimeError(f"attention block: a bounded hand-off wait gave up (code {code}); the decode steps since the last check are "
                               "invalid, the launch chain serves from here")

    def replay(self):
        self.graph.replay()

    def replay_steps(self, n):
        """n decode steps: whole multi-step graphs first, single-step graphs for the remainder."""
        k = self.steps_per_graph
        while self.graph_multi is not None and n >= k:
            self.graph_multi.replay()
            n -= k
        for _ in range(n):
            self.graph.replay()

    # -- accounting --------------------------------------------------------------
    def algorithmic_bytes_per_step(self, seq_len):
        cfg = self.model.cfg
        kvb = {"none": self.H * 2, "i8": self.H + 8, "u4": self.H // 2 + 8}[self.kv_mode]
        kv = len(self.model.layers) * self.B * 2 * self.g_loc * seq_len * kvb
        return self.model.weight_bytes + kv + getattr(self, "routed_expert_bytes", 0)

    def count_distinct_experts(self):
        """Mixture-of-experts models: one eager step on a copy of the state that records, per layer, how many DISTINCT local
        experts the batch selected -- each is streamed once per group of up to 4 slots that picked it, but algorithmically
        once per step.  Sets self.routed_expert_bytes (added to algorithmic_bytes_per_step) and returns the per-layer counts."""
        assert self.model.cfg.moe is not None
        ids0, old0, new0 = self.ids.clone(), self.old_lens.clone(), self.new_lens.clone()
        self._expert_log = []
        self.step()
        torch.cuda.synchronize()
        counts = [int((e[e >= 0]).unique().numel()) for e in self._expert_log]
        self._expert_log = None
        self.ids.copy_(ids0)
        self.old_lens.copy_(old0)
        self.new_lens.copy_(new0)
        self.routed_expert_bytes = sum(c * lw.expert_bytes for c, lw in zip(counts, self.model.layers))
        return counts
