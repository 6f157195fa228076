"""Tensor-level wrappers over the C-ABI (plumbing for tests, smoke and bench).

torch is used for device memory and streams only; every computation is a call into
lib/libdashinfer_hip.so through `capi`.  No CPU fallback exists here.
"""
import ctypes as C

import torch

from . import capi
from .capi import check, lib

_DT = {torch.float32: capi.F32, torch.float16: capi.F16, torch.bfloat16: capi.BF16}


def dt_code(t):
    return _DT[t.dtype if isinstance(t, torch.Tensor) else t]


def cur_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Scratch:
    """Shared scratch ("workspace" tensor of the allspark tensor map) + the zero-initialised
    arrival-counter buffer the split-K / split-KV kernels keep zero between launches."""

    def __init__(self, ws_bytes, device="cuda", sync_bytes=None):
        self.ws = torch.empty(max(int(ws_bytes), 256), dtype=torch.uint8, device=device)
        nsync = int(sync_bytes if sync_bytes is not None else lib().dihip_gemm_lowp_sync_bytes())
        self.sync = torch.zeros(nsync, dtype=torch.uint8, device=device)

    @property
    def ws_bytes(self):
        return self.ws.numel()


class PackedWeight:
    """A weight in dihip tile-major order (+ interleaved (scale, zero) words for wbits 4/8)."""

    def __init__(self, wbits, N, K, group, dtype, w, szp):
        self.wbits, self.N, self.K, self.group, self.dtype, self.w, self.sz = wbits, N, K, group, dtype, w, szp

    @property
    def nbytes(self):
        return self.w.numel() + (self.sz.numel() * 4 if self.sz is not None else 0)


def pack_lowp(wq, scales, zeros, group, wbits):
    """wq: int8 [K,N] (wbits 8) or uint8 [K,ceil(N/2)] (wbits 4) on the GPU; scales/zeros FT [G,N]."""
    K = wq.shape[0]
    N = scales.shape[-1]
    group = -1 if group in (None, 0, -1) else int(group)
    l = lib()
    w = torch.empty(l.dihip_gemm_lowp_packed_weight_bytes(wbits, N, K), dtype=torch.uint8, device=wq.device)
    szp = torch.empty(l.dihip_gemm_lowp_packed_sz_bytes(N, K, group) // 4, dtype=torch.int32, device=wq.device)
    check(l.dihip_gemm_lowp_pack(cur_stream(), wbits, ptr(wq.contiguous()), ptr(scales.contiguous()),
                                 ptr(zeros.contiguous()), N, K, group, dt_code(scales), ptr(w), ptr(szp)),
          "dihip_gemm_lowp_pack")
    return PackedWeight(wbits, N, K, group, scales.dtype, w, szp)


def pack_dense(w_kn):
    K, N = w_kn.shape
    l = lib()
    w = torch.empty(l.dihip_dense_packed_weight_bytes(N, K), dtype=torch.uint8, device=w_kn.device)
    check(l.dihip_dense_pack(cur_stream(), ptr(w_kn.contiguous()), N, K, dt_code(w_kn), ptr(w)), "dihip_dense_pack")
    return PackedWeight(16, N, K, -1, w_kn.dtype, w, None)


def lowp_workspace_bytes(wbits, M, N, K, group):
    return int(lib().dihip_gemm_lowp_workspace_bytes(wbits, M, N, K, -1 if group in (None, 0) else group))


def gemv_plan(wbits, M, N, K, group, dual=False):
    """Launch plan of the decode GEMV for a shape, or None when the general kernel serves it."""
    blocks, upb, wk, wn, lds = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    st = lib().dihip_debug_gemv_plan(wbits, M, N, K, -1 if group in (None, 0) else group, int(dual), C.byref(blocks),
                                     C.byref(upb), C.byref(wk), C.byref(wn), C.byref(lds))
    if st != 0:
        return None
    return {"blocks": blocks.value, "upb": upb.value, "WK": wk.value, "WN": wn.value, "lds_bytes": lds.value}


def gemm_lowp(x, pw, bias=None, residual=None, act=None, alpha=1.0, scratch=None, use_sync=True, out=None):
    """op GemmA16W8 / GemmA16W4: x FT [..., K] -> FT [..., N]."""
    M = x.numel() // pw.K
    y = out if out is not None else torch.empty(*x.shape[:-1], pw.N, dtype=x.dtype, device=x.device)
    if scratch is None:
        scratch = Scratch(lowp_workspace_bytes(pw.wbits, max(M, 1), pw.N, pw.K, pw.group), x.device)
    fn = lib().dihip_gemm_a16w8 if pw.wbits == 8 else lib().dihip_gemm_a16w4
    check(fn(cur_stream(), ptr(x), ptr(pw.w), ptr(pw.sz), ptr(bias), ptr(residual), ptr(y), M, pw.N, pw.K, pw.group,
             capi.ACT[act], float(alpha), ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync) if use_sync else None,
             dt_code(x)), "dihip_gemm_a16wX")
    return y


def gemm_dense(x, pw, bias=None, residual=None, act=None, alpha=1.0, scratch=None, out=None):
    M = x.numel() // pw.K
    y = out if out is not None else torch.empty(*x.shape[:-1], pw.N, dtype=x.dtype, device=x.device)
    if scratch is None:
        scratch = Scratch(lib().dihip_dense_workspace_bytes(max(M, 1), pw.N, pw.K), x.device)
    check(lib().dihip_gemm_a16w16(cur_stream(), ptr(x), ptr(pw.w), ptr(bias), ptr(residual), ptr(y), M, pw.N, pw.K,
                                  capi.ACT[act], float(alpha), ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync),
                                  dt_code(x)), "dihip_gemm_a16w16")
    return y


def fused_norm_gemm(h, gamma, eps, pw, bias, scratch, act=None, out=None):
    M = h.shape[0]
    y = out if out is not None else torch.empty(M, pw.N, dtype=pw.dtype, device=h.device)
    check(lib().dihip_fused_norm_gemm(cur_stream(), pw.wbits, ptr(h), ptr(gamma), float(eps), ptr(pw.w), ptr(pw.sz),
                                      ptr(bias), ptr(y), M, pw.N, pw.K, pw.group, capi.ACT[act], ptr(scratch.ws),
                                      scratch.ws_bytes, ptr(scratch.sync), dt_code(pw.dtype)), "dihip_fused_norm_gemm")
    return y


# ---------------------------------------------------------------------------- mixture of experts
class PackedExperts:
    """num_experts equally shaped quantised matrices [K, N], each packed like PackedWeight, stacked back to back."""

    def __init__(self, w, sz, wbits, N, K, group, num_experts):
        self.w, self.sz, self.wbits, self.N, self.K, self.group, self.num_experts = w, sz, wbits, N, K, group, num_experts


def pack_experts(qs, scales, zeros, group, wbits):
    """qs / scales / zeros: lists (one entry per expert) in the layout pack_lowp takes."""
    E = len(qs)
    packed = [pack_lowp(q.contiguous(), s_.contiguous(), z.contiguous(), group, wbits) for q, s_, z in zip(qs, scales, zeros)]
    p0 = packed[0]
    wb = int(lib().dihip_gemm_lowp_packed_weight_bytes(wbits, p0.N, p0.K))
    sb = int(lib().dihip_gemm_lowp_packed_sz_bytes(p0.N, p0.K, p0.group))
    assert p0.w.numel() * p0.w.element_size() == wb and p0.sz.numel() * p0.sz.element_size() == sb
    w = torch.cat([p.w.view(torch.uint8).reshape(-1) for p in packed])
    sz = torch.cat([p.sz.view(torch.uint8).reshape(-1) for p in packed])
    return PackedExperts(w, sz, wbits, p0.N, p0.K, p0.group, E)


def moe_route(logits, top_k, ep=None, scores=None, experts=None):
    """ep = (first, count): expert-parallel window of this rank (indices come out local to it, -1 elsewhere)."""
    T, E = logits.shape
    scores = scores if scores is not None else torch.empty(T, top_k, dtype=torch.float32, device=logits.device)
    experts = experts if experts is not None else torch.empty(T, top_k, dtype=torch.int32, device=logits.device)
    first, count = ep if ep is not None else (0, E)
    check(lib().dihip_moe_route_ep(cur_stream(), ptr(logits), T, E, top_k, ptr(scores), ptr(experts), dt_code(logits), first, count),
          "dihip_moe_route_ep")
    return scores, experts


def moe_shared_combine(h_out, h_res, moe_out, shared_out, shared_gate):
    T, hidden = moe_out.shape
    check(lib().dihip_moe_shared_combine(cur_stream(), ptr(h_out), ptr(h_res), ptr(moe_out), ptr(shared_out), ptr(shared_gate), T, hidden,
                                         dt_code(moe_out)), "dihip_moe_shared_combine")
    return h_out


MOE_PREGROUPED, MOE_NO_FINALIZE = 1, 2
MOE_GROUP_MAX_SLOTS = 2048


def moe_experts(x, experts, scores, gate, up, down, ws=None, out=None, flags=0):
    """flags: MOE_PREGROUPED (the group tables in `ws` come from moe_route_grouped), MOE_NO_FINALIZE (the slot outputs stay in
    `ws` for moe_combine; nothing is written to `out`)."""
    T, hidden = x.shape
    top_k = experts.shape[1]
    proj = gate.N
    need = int(lib().dihip_moe_workspace_bytes(T, top_k, hidden, proj))
    ws = ws if ws is not None else torch.empty(need, dtype=torch.uint8, device=x.device)
    if not flags & MOE_NO_FINALIZE:
        out = out if out is not None else torch.empty(T, hidden, dtype=x.dtype, device=x.device)
    check(lib().dihip_moe_experts_ex(cur_stream(), gate.wbits, ptr(x), ptr(experts), ptr(scores), ptr(gate.w), ptr(gate.sz), ptr(up.w),
                                     ptr(up.sz), ptr(down.w), ptr(down.sz), T, top_k, hidden, proj, gate.group, ptr(out), ptr(ws),
                                     ws.numel(), dt_code(x), int(flags)), "dihip_moe_experts_ex")
    return out


def moe_route_grouped(logits, top_k, hidden, proj, ws, ep=None, scores=None, experts=None):
    """moe_route + the slot grouping of moe_experts in one launch (decode batches: 1 < T, T * top_k <= MOE_GROUP_MAX_SLOTS);
    follow with moe_experts(..., ws=ws, flags=MOE_PREGROUPED)."""
    T, E = logits.shape
    scores = scores if scores is not None else torch.empty(T, top_k, dtype=torch.float32, device=logits.device)
    experts = experts if experts is not None else torch.empty(T, top_k, dtype=torch.int32, device=logits.device)
    first, count = ep if ep is not None else (0, E)
    check(lib().dihip_moe_route_grouped(cur_stream(), ptr(logits), T, E, top_k, ptr(scores), ptr(experts), dt_code(logits), first, count,
                                        hidden, proj, ptr(ws), ws.numel()), "dihip_moe_route_grouped")
    return scores, experts


def moe_router_gate(xn, router, gate, logits=None, sig=None):
    """The router Gemm (xn . W_router -> FT [T, E]) and the shared expert's gate Gemm with SIGMOID (-> FT [T, 1]) in one launch;
    router / gate: pack_dense of [hidden, E] / [hidden, 1]."""
    T, hidden = xn.shape
    logits = logits if logits is not None else torch.empty(T, router.N, dtype=xn.dtype, device=xn.device)
    sig = sig if sig is not None else torch.empty(T, 1, dtype=xn.dtype, device=xn.device)
    assert router.K == hidden and gate.K == hidden and gate.N == 1
    check(lib().dihip_moe_router_gate(cur_stream(), ptr(xn), ptr(router.w), ptr(gate.w), ptr(logits), ptr(sig), T, router.N, hidden,
                                      dt_code(xn)), "dihip_moe_router_gate")
    return logits, sig


def moe_combine(h_out, h_res, ws, scores, experts, shared_out, shared_gate, proj):
    """finalize-routing over the slot outputs moe_experts(..., flags=MOE_NO_FINALIZE) left in `ws` + moe_shared_combine's tail."""
    T, hidden = shared_out.shape
    check(lib().dihip_moe_combine(cur_stream(), ptr(h_out), ptr(h_res), ptr(ws), ptr(scores), ptr(experts), ptr(shared_out),
                                  ptr(shared_gate), T, experts.shape[1], hidden, proj, dt_code(shared_out)), "dihip_moe_combine")
    return h_out


ACT_ROWMAJOR, ACT_FRAG32 = 0, 1


def prefers_frag(pw, M, dual=False):
    """True when a [M, pw.N, pw.K] call runs on the small-batch kernel, which reads / writes the FRAG32
    activation layout (include/dashinfer_hip.h) faster than row-major."""
    if pw.dtype != torch.bfloat16:   # the small-batch kernels (and with them the FRAG32 layout) are bf16; f16 takes the general kernel
        return False
    return bool(lib().dihip_gemm_lowp_prefers_frag(pw.wbits, int(M), pw.N, pw.K, pw.group, int(dual)))


def act_frag_numel(M, K):
    return int(lib().dihip_act_frag_bytes(int(M), int(K))) // 2


def act_to_frag(x):
    M, K = x.shape
    out = torch.zeros(act_frag_numel(M, K), dtype=x.dtype, device=x.device)
    check(lib().dihip_act_to_frag(cur_stream(), ptr(x), ptr(out), M, K, dt_code(x)), "dihip_act_to_frag")
    return out


def act_from_frag(xf, M, K):
    out = torch.empty(M, K, dtype=xf.dtype, device=xf.device)
    check(lib().dihip_act_from_frag(cur_stream(), ptr(xf), ptr(out), M, K, dt_code(xf)), "dihip_act_from_frag")
    return out


def fused_norm_swiglu(h, gamma, eps, pg, pu, scratch, out=None, y_layout=ACT_ROWMAJOR):
    M = h.shape[0]
    if out is not None:
        y = out
    elif y_layout == ACT_FRAG32:
        y = torch.zeros(act_frag_numel(M, pg.N), dtype=pg.dtype, device=h.device)
    else:
        y = torch.empty(M, pg.N, dtype=pg.dtype, device=h.device)
    check(lib().dihip_fused_norm_swiglu_ex(cur_stream(), pg.wbits, ptr(h), ptr(gamma), float(eps), ptr(pg.w), ptr(pg.sz),
                                           ptr(pu.w), ptr(pu.sz), ptr(y), M, pg.N, pg.K, pg.group, ptr(scratch.ws),
                                           scratch.ws_bytes, ptr(scratch.sync), dt_code(pg.dtype), int(y_layout)),
          "dihip_fused_norm_swiglu_ex")
    return y


def fused_gemm_addto(x, pw, h_res, scratch, out=None, x_layout=ACT_ROWMAJOR, M=None):
    M = x.shape[0] if M is None else M
    h_out = out if out is not None else torch.empty(M, pw.N, dtype=torch.float32, device=x.device)
    check(lib().dihip_fused_gemm_addto_ex(cur_stream(), pw.wbits, ptr(x), ptr(pw.w), ptr(pw.sz), ptr(h_res), ptr(h_out),
                                          M, pw.N, pw.K, pw.group, ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync),
                                          dt_code(x), int(x_layout)), "dihip_fused_gemm_addto_ex")
    return h_out


def fused_gemm_addto_norm(x, pw, h_res, scratch, gamma, eps, xnorm, out=None, x_layout=ACT_ROWMAJOR, xnorm_layout=ACT_ROWMAJOR, M=None):
    """h_out = h_res + x.W and xnorm = RMSNorm(h_out) (FT; row-major or FRAG32) in one call: the norm rides on the split-K
    reduction when the plan has one."""
    M = x.shape[0] if M is None else M
    h_out = out if out is not None else torch.empty(M, pw.N, dtype=torch.float32, device=x.device)
    check(lib().dihip_fused_gemm_addto_norm(cur_stream(), pw.wbits, ptr(x), ptr(pw.w), ptr(pw.sz), ptr(h_res), ptr(h_out),
                                            M, pw.N, pw.K, pw.group, ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync),
                                            dt_code(x), int(x_layout), ptr(gamma), float(eps), ptr(xnorm), int(xnorm_layout)),
          "dihip_fused_gemm_addto_norm")
    return h_out


def prenorm_gemm(xnorm, pw, bias, scratch, M, x_layout=ACT_ROWMAJOR, act=None, out=None):
    y = out if out is not None else torch.empty(M, pw.N, dtype=pw.dtype, device=xnorm.device)
    check(lib().dihip_prenorm_gemm(cur_stream(), pw.wbits, ptr(xnorm), int(x_layout), ptr(pw.w), ptr(pw.sz), ptr(bias), ptr(y),
                                   M, pw.N, pw.K, pw.group, capi.ACT[act], ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync),
                                   dt_code(pw.dtype)), "dihip_prenorm_gemm")
    return y


def prenorm_swiglu(xnorm, pg, pu, scratch, M, x_layout=ACT_ROWMAJOR, out=None, y_layout=ACT_ROWMAJOR):
    if out is None:
        out = (torch.zeros(act_frag_numel(M, pg.N), dtype=pg.dtype, device=xnorm.device) if y_layout == ACT_FRAG32
               else torch.empty(M, pg.N, dtype=pg.dtype, device=xnorm.device))
    check(lib().dihip_prenorm_swiglu(cur_stream(), pg.wbits, ptr(xnorm), int(x_layout), ptr(pg.w), ptr(pg.sz), ptr(pu.w), ptr(pu.sz),
                                     ptr(out), M, pg.N, pg.K, pg.group, ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync),
                                     dt_code(pg.dtype), int(y_layout)), "dihip_prenorm_swiglu")
    return out


# -- the same three with the RMSNorm DEFERRED (include/dashinfer_hip.h, "with the RMSNorm DEFERRED"): the producer leaves
# FT(gamma * h) and per-workgroup partial sums of h^2, the consumer scales its accumulators by 1 / rms
def rowsq_buffer(device):
    return torch.zeros(int(lib().dihip_rowsq_bytes()) // 4, dtype=torch.float32, device=device)


def prenorm_rowsq_supported(pw, M, dual=False, x_layout=ACT_ROWMAJOR):
    return bool(lib().dihip_prenorm_rowsq_supported(pw.wbits, int(M), pw.N, pw.K, pw.group, 1 if dual else 0, dt_code(pw.dtype), int(x_layout)))


def fused_gemm_addto_prenorm(x, pw, h_res, scratch, gamma, eps, xnorm, rowsq, out=None, x_layout=ACT_ROWMAJOR, xnorm_layout=ACT_ROWMAJOR, M=None):
    """-> (h_out, parts): parts > 0 -- xnorm = FT(gamma * h_out), rowsq[part][32] partial sums of h_out^2; parts == 0 -- xnorm is
    the finished norm (the kernel that served the call does not offer the deferred form)"""
    import ctypes as _C
    M = x.shape[0] if M is None else M
    h_out = out if out is not None else torch.empty(M, pw.N, dtype=torch.float32, device=x.device)
    parts = _C.c_int(0)
    check(lib().dihip_fused_gemm_addto_prenorm(cur_stream(), pw.wbits, ptr(x), ptr(pw.w), ptr(pw.sz), ptr(h_res), ptr(h_out),
                                               M, pw.N, pw.K, pw.group, ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync),
                                               dt_code(x), int(x_layout), ptr(gamma), float(eps), ptr(xnorm), int(xnorm_layout),
                                               ptr(rowsq), rowsq.numel() * 4, _C.addressof(parts)), "dihip_fused_gemm_addto_prenorm")
    return h_out, int(parts.value)


def prenorm_gemm_rowsq(xnorm, pw, bias, scratch, M, rowsq, parts, eps, x_layout=ACT_ROWMAJOR, act=None, out=None):
    y = out if out is not None else torch.empty(M, pw.N, dtype=pw.dtype, device=xnorm.device)
    check(lib().dihip_prenorm_gemm_rowsq(cur_stream(), pw.wbits, ptr(xnorm), int(x_layout), ptr(pw.w), ptr(pw.sz), ptr(bias), ptr(y),
                                         M, pw.N, pw.K, pw.group, capi.ACT[act], ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync),
                                         dt_code(pw.dtype), ptr(rowsq), int(parts), float(eps)), "dihip_prenorm_gemm_rowsq")
    return y


def prenorm_swiglu_rowsq(xnorm, pg, pu, scratch, M, rowsq, parts, eps, x_layout=ACT_ROWMAJOR, out=None, y_layout=ACT_ROWMAJOR):
    if out is None:
        out = (torch.zeros(act_frag_numel(M, pg.N), dtype=pg.dtype, device=xnorm.device) if y_layout == ACT_FRAG32
               else torch.empty(M, pg.N, dtype=pg.dtype, device=xnorm.device))
    check(lib().dihip_prenorm_swiglu_rowsq(cur_stream(), pg.wbits, ptr(xnorm), int(x_layout), ptr(pg.w), ptr(pg.sz), ptr(pu.w), ptr(pu.sz),
                                           ptr(out), M, pg.N, pg.K, pg.group, ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync),
                                           dt_code(pg.dtype), int(y_layout), ptr(rowsq), int(parts), float(eps)), "dihip_prenorm_swiglu_rowsq")
    return out


def lm_head(h, gamma, eps, pw, scratch, out=None):
    M = h.shape[0]
    logits = out if out is not None else torch.empty(M, pw.N, dtype=torch.float32, device=h.device)
    check(lib().dihip_lm_head(cur_stream(), ptr(logits), ptr(h), ptr(gamma), float(eps), ptr(pw.w), M, pw.N, pw.K,
                              ptr(scratch.ws), scratch.ws_bytes, ptr(scratch.sync), dt_code(pw.dtype)), "dihip_lm_head")
    return logits


# ------------------------------------------------------------------------------ KV spans ----
def span_bytes(g, S, H, mode, dtype):
    return int(lib().dihip_span_bytes(g, S, H, capi.KV[mode], dt_code(dtype)))


class SpanPool:
    """Device pool of KV spans handed out in a strided (non-contiguous) order, like the
    reference's span-attention test (test_quant_none.cpp:479-506).  Stands in for the
    reference's CacheFrameManager / CacheSpanManager (out of scope, host code)."""

    def __init__(self, num_spans, g, S, H, mode, dtype, device="cuda", stride=7):
        self.nbytes = span_bytes(g, S, H, mode, dtype)
        self.aligned = (self.nbytes + 255) // 256 * 256
        self.pool = torch.zeros(num_spans * self.aligned, dtype=torch.uint8, device=device)
        self.g, self.S, self.H, self.mode, self.dtype = g, S, H, mode, dtype
        from math import gcd
        while gcd(stride, num_spans) != 1:
            stride += 1
        order = [(i * stride) % num_spans for i in range(num_spans)]
        self.free = order

    def alloc(self):
        idx = self.free.pop(0)
        return self.pool.data_ptr() + idx * self.aligned, idx

    def span_view(self, idx):
        return self.pool[idx * self.aligned: idx * self.aligned + self.nbytes]


class KVCacheSet:
    """Per-batch span pointer tables for K and V ([B][span_stride] device int64) for one layer."""

    def __init__(self, pool, batch, max_spans):
        self.pool, self.batch, self.max_spans = pool, batch, max_spans
        self.k_idx = [[] for _ in range(batch)]
        self.v_idx = [[] for _ in range(batch)]
        self.k_host = torch.zeros(batch, max_spans, dtype=torch.int64)
        self.v_host = torch.zeros(batch, max_spans, dtype=torch.int64)
        self.k_ptrs = torch.zeros(batch, max_spans, dtype=torch.int64, device=pool.pool.device)
        self.v_ptrs = torch.zeros(batch, max_spans, dtype=torch.int64, device=pool.pool.device)

    def ensure(self, b, ntokens):
        S = self.pool.S
        changed = False
        while len(self.k_idx[b]) * S < ntokens:
            kp, ki = self.pool.alloc()
            vp, vi = self.pool.alloc()
            self.k_host[b, len(self.k_idx[b])] = kp
            self.v_host[b, len(self.v_idx[b])] = vp
            self.k_idx[b].append(ki)
            self.v_idx[b].append(vi)
            changed = True
        return changed

    def sync(self):
        self.k_ptrs.copy_(self.k_host)
        self.v_ptrs.copy_(self.v_host)


def kv_append(kv, q_out, qkv, old_lens, n, g, H):
    pool = kv.pool
    B = qkv.shape[0]
    check(lib().dihip_kv_append(cur_stream(), ptr(kv.k_ptrs), ptr(kv.v_ptrs), ptr(q_out), ptr(qkv), ptr(old_lens), B, n,
                                g, H, pool.S, kv.max_spans, capi.KV[pool.mode], dt_code(qkv)), "dihip_kv_append")


def rope_kv_append(kv, q_out, qkv, old_lens, inv_freq, n, g, H):
    pool = kv.pool
    B = qkv.shape[0]
    check(lib().dihip_rope_kv_append(cur_stream(), ptr(kv.k_ptrs), ptr(kv.v_ptrs), ptr(q_out), ptr(qkv), ptr(old_lens),
                                     ptr(inv_freq), B, n, g, H, pool.S, kv.max_spans, capi.KV[pool.mode], dt_code(qkv)),
          "dihip_rope_kv_append")


def kv_context_copy(span_ptrs_row, src, src_stride, seq_len, start_pos, g, H, S, mode):
    check(lib().dihip_kv_context_copy(cur_stream(), ptr(span_ptrs_row), ptr(src), src_stride, seq_len, start_pos, g, H,
                                      S, capi.KV[mode], dt_code(src)), "dihip_kv_context_copy")


def kv_prefix_gather(dst, span_ptrs_row, prefix_len, g, H, S, mode):
    check(lib().dihip_kv_prefix_gather(cur_stream(), ptr(dst), ptr(span_ptrs_row), prefix_len, g, H, S, capi.KV[mode],
                                       dt_code(dst)), "dihip_kv_prefix_gather")


def span_attn_workspace(batch, n, H, max_len):
    return int(lib().dihip_span_attn_decode_workspace_bytes(batch, n, H, max_len, 0))


def span_attn_decode(q, kv, seq_lens_dev, n, g, H, max_len, scale, ws, sync, out=None, out_layout=0):
    """out_layout = ACT_FRAG32: the [B, n*H] result in the MFMA-fragment layout the small-batch o-projection reads."""
    B = q.shape[0]
    if out is None:
        out = (torch.zeros(act_frag_numel(B, n * H), dtype=q.dtype, device=q.device) if out_layout
               else torch.empty(B, n * H, dtype=q.dtype, device=q.device))
    pool = kv.pool
    check(lib().dihip_span_attn_decode_sync(cur_stream(), ptr(out), ptr(q), ptr(kv.k_ptrs), ptr(kv.v_ptrs), ptr(seq_lens_dev),
                                            B, n, g, H, pool.S, kv.max_spans, max_len, capi.KV[pool.mode], dt_code(q),
                                            float(scale), ptr(ws), ws.numel() if ws is not None else 0, ptr(sync),
                                            sync.numel() if sync is not None else 0, int(out_layout)), "dihip_span_attn_decode_sync")
    return out


def prefill_attn(q, k, v, n, g, H, alpha, causal=True, out=None):
    """Causal GQA prefill attention.  q: FT [Lq, >= n*H] (row stride = q.stride(0)), k / v: FT
    [Lk, >= g*H] views of contiguous (MIX) or fused-qkv (INTERLEAVED) rows -> FT [Lq, n*H]."""
    Lq, Lk = q.shape[0], k.shape[0]
    out = out if out is not None else torch.empty(Lq, n * H, dtype=q.dtype, device=q.device)
    assert k.stride(0) == v.stride(0)
    check(lib().dihip_prefill_attn(cur_stream(), ptr(out), ptr(q), ptr(k), ptr(v), Lq, Lk, q.stride(0), k.stride(0), n, g, H,
                                   1 if causal else 0, float(alpha), dt_code(q)), "dihip_prefill_attn")
    return out


def rope_table(inv_freq, max_pos, H):
    tab = torch.empty(max_pos, H // 2, 2, dtype=torch.float32, device=inv_freq.device)
    check(lib().dihip_rope_table(cur_stream(), ptr(tab), ptr(inv_freq), max_pos, H), "dihip_rope_table")
    return tab


def span_attn_fused_workspace(batch, n, g, H, max_len):
    return int(lib().dihip_span_attn_fused_workspace_bytes(batch, n, g, H, max_len))


def span_attn_decode_fused(qkv, kv, old_lens_dev, rope_tab, n, g, H, max_len, scale, ws, out=None, sync=None):
    """Rotary + cache append + paged decode attention of one step from the fused qkv rows.  sync (zeroed once,
    dihip_span_attn_sync_bytes): the split partials are merged inside the launch instead of by a second one."""
    B = qkv.shape[0]
    out = out if out is not None else torch.empty(B, n * H, dtype=qkv.dtype, device=qkv.device)
    pool = kv.pool
    if sync is not None:
        check(lib().dihip_span_attn_decode_fused_sync(cur_stream(), ptr(out), ptr(qkv), ptr(kv.k_ptrs), ptr(kv.v_ptrs), ptr(old_lens_dev),
                                                      ptr(rope_tab), B, n, g, H, pool.S, kv.max_spans, max_len, capi.KV[pool.mode],
                                                      dt_code(qkv), float(scale), ptr(ws), ws.numel(), ptr(sync), sync.numel()),
              "dihip_span_attn_decode_fused_sync")
        return out
    check(lib().dihip_span_attn_decode_fused(cur_stream(), ptr(out), ptr(qkv), ptr(kv.k_ptrs), ptr(kv.v_ptrs), ptr(old_lens_dev),
                                             ptr(rope_tab), B, n, g, H, pool.S, kv.max_spans, max_len, capi.KV[pool.mode],
                                             dt_code(qkv), float(scale), ptr(ws), ws.numel()), "dihip_span_attn_decode_fused")
    return out


def span_attn_decode_step(qkv, kv, old_lens_dev, rope_tab, n, g, H, max_len, scale, ws, sync, out=None, out_layout=ACT_ROWMAJOR):
    """span_attn_decode_fused with the output layout of span_attn_decode: the uint4 cache (bf16 rows) is one launch too --
    Rotary, the quantising append and the attention -- and may write the o projection's FRAG32 layout."""
    B = qkv.shape[0]
    if out is None:
        out = (torch.zeros(act_frag_numel(B, n * H), dtype=qkv.dtype, device=qkv.device) if out_layout == ACT_FRAG32
               else torch.empty(B, n * H, dtype=qkv.dtype, device=qkv.device))
    pool = kv.pool
    check(lib().dihip_span_attn_decode_step(cur_stream(), ptr(out), ptr(qkv), ptr(kv.k_ptrs), ptr(kv.v_ptrs), ptr(old_lens_dev), ptr(rope_tab),
                                            B, n, g, H, pool.S, kv.max_spans, max_len, capi.KV[pool.mode], dt_code(qkv), float(scale),
                                            ptr(ws), ws.numel(), ptr(sync), sync.numel() if sync is not None else 0, int(out_layout)),
          "dihip_span_attn_decode_step")
    return out


def decode_attn_block_supported(qkv_w, hidden, n, g, H, max_len, kv_mode, dtype, batch):
    """True when dihip_decode_attn_block serves this configuration (batch 1, bf16, int4 g128 or int8 per channel, 16-bit cache, ...)."""
    return bool(lib().dihip_decode_attn_block_supported(qkv_w.wbits, qkv_w.group, hidden, n, g, H, max_len, capi.KV[kv_mode],
                                                         capi.BF16 if dtype == torch.bfloat16 else capi.F16, batch))


def decode_attn_block(h, h_res, gamma, eps, qkv_w, qkv_bias, o_w, kv, old_lens_dev, rope_tab, n, g, H, max_len, scale, ws, sync, out=None):
    """RMSNorm + qkv GEMV, Rotary + cache append + paged attention (+ split merge) and the o-projection + residual of ONE
    request in ONE launch (dihip_decode_attn_block): out = h_res + attention(norm(h)) . Wo, bit-identical to the three calls.
    h: f32 [1, hidden]; sync: zeroed once (dihip_decode_attn_block_sync_bytes); ws: dihip_decode_attn_block_workspace_bytes."""
    out = out if out is not None else torch.empty_like(h)
    pool = kv.pool
    check(lib().dihip_decode_attn_block(cur_stream(), qkv_w.wbits, ptr(h), ptr(h_res) if h_res is not None else None, ptr(out), ptr(gamma),
                                        float(eps), ptr(qkv_w.w), ptr(qkv_w.sz), ptr(qkv_bias) if qkv_bias is not None else None,
                                        ptr(o_w.w), ptr(o_w.sz), ptr(kv.k_ptrs), ptr(kv.v_ptrs), ptr(old_lens_dev), ptr(rope_tab),
                                        h.shape[-1], n, g, H, qkv_w.group, pool.S, kv.max_spans, max_len, capi.KV[pool.mode],
                                        dt_code(gamma), float(scale), ptr(ws), ws.numel(), ptr(sync), sync.numel()),
          "dihip_decode_attn_block")
    return out


def decode_mlp_block_supported(gate_w, hidden, dtype, batch):
    """True when dihip_decode_mlp_block serves this configuration (batch 1, bf16, int4 g128, decode-GEMV shapes)."""
    return bool(lib().dihip_decode_mlp_block_supported(gate_w.wbits, gate_w.group, hidden, gate_w.N, capi.BF16 if dtype == torch.bfloat16 else capi.F16,
                                                        batch))


def decode_mlp_block(h, h_res, gamma, eps, gate_w, up_w, down_w, sync, out=None):
    """RMSNorm + gate / up GEMV + SwiGLU and the down projection + residual of ONE request in ONE launch (dihip_decode_mlp_block):
    out = h_res + SwiGLU(norm(h)) . Wdown, bit-identical to fused_norm_swiglu + fused_gemm_addto.  sync: zeroed once."""
    out = out if out is not None else torch.empty_like(h)
    check(lib().dihip_decode_mlp_block(cur_stream(), gate_w.wbits, ptr(h), ptr(h_res) if h_res is not None else None, ptr(out), ptr(gamma), float(eps),
                                       ptr(gate_w.w), ptr(gate_w.sz), ptr(up_w.w), ptr(up_w.sz), ptr(down_w.w), ptr(down_w.sz), h.shape[-1],
                                       gate_w.N, gate_w.group, dt_code(gamma), ptr(sync), sync.numel()), "dihip_decode_mlp_block")
    return out


def span_attn_merge_partials(partials, batch, n, nsplits, dtype=torch.bfloat16):
    out = torch.empty(batch, n * 128, dtype=dtype, device=partials.device)
    check(lib().dihip_span_attn_merge_partials(cur_stream(), ptr(out), ptr(partials), batch, n, nsplits, dt_code(dtype)),
          "dihip_span_attn_merge_partials")
    return out


def rmsnorm(x, gamma, eps):
    y = torch.empty_like(x)
    check(lib().dihip_rmsnorm(cur_stream(), ptr(y), ptr(x), ptr(gamma), float(eps), x.numel() // x.shape[-1],
                              x.shape[-1], dt_code(x)), "dihip_rmsnorm")
    return y


def rmsnorm_rows(h, gamma, eps, out=None):
    """f32 hidden rows -> FT normalised rows (the norm the fused GEMV entries apply, as its own launch)."""
    M, K = h.shape
    out = out if out is not None else torch.empty(M, K, dtype=gamma.dtype, device=h.device)
    check(lib().dihip_rmsnorm_rows(cur_stream(), ptr(out), ptr(h), ptr(gamma), float(eps), M, K, dt_code(gamma)), "dihip_rmsnorm_rows")
    return out


def rope_qk_(qkv, positions, inv_freq, n, g, H):
    check(lib().dihip_rope_qk(cur_stream(), ptr(qkv), ptr(positions), ptr(inv_freq), qkv.shape[0], n, g, H,
                              dt_code(qkv)), "dihip_rope_qk")
    return qkv


def binary_add(a, b):
    y = torch.empty_like(a)
    check(lib().dihip_binary_add(cur_stream(), ptr(y), ptr(a), ptr(b), a.numel(), dt_code(a)), "dihip_binary_add")
    return y


def silu_mul(gate, up):
    y = torch.empty_like(gate)
    check(lib().dihip_silu_mul(cur_stream(), ptr(y), ptr(gate), ptr(up), gate.numel(), dt_code(gate)), "dihip_silu_mul")
    return y


def argmax(logits, ws=None, out=None, advance=()):
    """Greedy sampling; `advance`: up to two per-request u32 counter tensors incremented in the same launch."""
    M, N = logits.shape
    ids = out if out is not None else torch.empty(M, dtype=torch.int64, device=logits.device)
    ws = ws if ws is not None else torch.empty(M * 64 * 8, dtype=torch.uint8, device=logits.device)
    if advance:
        a = advance[0]
        b = advance[1] if len(advance) > 1 else None
        check(lib().dihip_argmax_advance(cur_stream(), ptr(ids), ptr(logits), M, N, ptr(ws), ws.numel(), ptr(a), ptr(b)),
              "dihip_argmax_advance")
    else:
        check(lib().dihip_argmax(cur_stream(), ptr(ids), ptr(logits), M, N, ptr(ws), ws.numel()), "dihip_argmax")
    return ids


def embedding(ids, table, out=None):
    M, K = ids.shape[0], table.shape[1]
    h = out if out is not None else torch.empty(M, K, dtype=torch.float32, device=table.device)
    check(lib().dihip_embedding_v(cur_stream(), ptr(h), ptr(ids), ptr(table), M, K, table.shape[0], dt_code(table)), "dihip_embedding_v")
    return h


def prefetch(tensors, stream=None, workgroups=0):
    """Enqueue a read-only sweep of up to 8 device tensors (Infinity Cache prefetch) on `stream`."""
    tensors = [t for t in tensors if t is not None]
    n = len(tensors)
    bufs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    sizes = (C.c_size_t * n)(*[t.numel() * t.element_size() for t in tensors])
    st = cur_stream() if stream is None else C.c_void_p(stream.cuda_stream)
    check(lib().dihip_prefetch(st, bufs, sizes, n, workgroups), "dihip_prefetch")


def increment_u32_(v):
    check(lib().dihip_increment_u32(cur_stream(), ptr(v), v.numel()), "dihip_increment_u32")
    return v


def sample(logits, top_k, top_p, temperature, seed, position=None, advance=(), want_probs=False):
    """The sampling half of GenerateOp (dihip_sample): logits f32 [M, N]; top_k / top_p / temperature / seed per row (lists or
    tensors); position: u32 device tensor [M] (the index of the token being sampled) or None.  -> ids [M] (and, with want_probs,
    the final probabilities / candidate indices [M, 1024] of the sorted candidates)."""
    M, N = logits.shape
    dev = logits.device
    tk = torch.as_tensor(top_k, dtype=torch.int32).to(dev)
    tp = torch.as_tensor(top_p, dtype=torch.float32).to(dev)
    tt = torch.as_tensor(temperature, dtype=torch.float32).to(dev)
    sd = torch.as_tensor([int(s) & 0x7FFFFFFFFFFFFFFF for s in seed] if not isinstance(seed, torch.Tensor) else seed, dtype=torch.int64).to(dev)
    ids = torch.empty(M, dtype=torch.int64, device=dev)
    probs = torch.zeros(M, 1024, dtype=torch.float32, device=dev) if want_probs else None
    cand = torch.full((M, 1024), -1, dtype=torch.int32, device=dev) if want_probs else None
    a = advance[0] if len(advance) > 0 else None
    b = advance[1] if len(advance) > 1 else None
    check(lib().dihip_sample(cur_stream(), ptr(ids), ptr(logits), M, N, ptr(tk), ptr(tp), ptr(tt), ptr(sd), ptr(position), ptr(a), ptr(b),
                             ptr(probs), ptr(cand)), "dihip_sample")
    return (ids, probs, cand) if want_probs else ids


def logits_processor_(logits, ids, cur_len, input_len, repetition_penalty=None, frequency_penalty=None, presence_penalty=None,
                      no_repeat_ngram_size=None, min_length=None, eos_token_id=None, suppress_repetition_in_generation=None, ws=None):
    """GenerateOp's logits processors, in place on f32 logits [M, N] (dihip_logits_processor; cuda::LogitsProcessor,
    csrc/core/kernel/cuda/beam_search.cu:456-539).  ids: int64 [M, max_len] device tensor; the per-request lists are python lists or
    tensors of length M (None: the processor's neutral value)."""
    M, N = logits.shape
    dev = logits.device
    assert logits.dtype == torch.float32 and logits.is_contiguous() and ids.dtype == torch.int64 and ids.is_contiguous() and ids.shape[0] == M

    def lst(v, dt, default):
        return torch.as_tensor([default] * M if v is None else v, dtype=dt).to(dev)
    cl, il = lst(cur_len, torch.int32, 0), lst(input_len, torch.int32, 0)
    rp, fq, pr = lst(repetition_penalty, torch.float32, 1.0), lst(frequency_penalty, torch.float32, 0.0), lst(presence_penalty, torch.float32, 0.0)
    ng, ml, eos = lst(no_repeat_ngram_size, torch.int32, 0), lst(min_length, torch.int32, 0), lst(eos_token_id, torch.int32, -1)
    sup = lst(suppress_repetition_in_generation, torch.int32, 0)
    need = lib().dihip_logits_processor_workspace_bytes(M, N)
    if ws is None:
        ws = torch.empty(max(need, 4), dtype=torch.uint8, device=dev)
    check(lib().dihip_logits_processor(cur_stream(), ptr(logits), M, N, ptr(ids), ids.shape[1], ptr(cl), ptr(il), ptr(rp), ptr(fq), ptr(pr), ptr(ng),
                                       ptr(ml), ptr(eos), ptr(sup), ptr(ws), ws.numel()), "dihip_logits_processor")
    return logits


def logprobs(logits, chosen=None, top_n=0):
    """GenerateOp's log-probability outputs (dihip_logprobs; generate_impl_gpu.hpp:33-80): -> (token_logprob [M] or None, top values
    [M, top_n] f32, top indices [M, top_n] i32)."""
    M, N = logits.shape
    dev = logits.device
    assert logits.dtype == torch.float32 and logits.is_contiguous()
    tok = torch.empty(M, dtype=torch.float32, device=dev) if chosen is not None else None
    tv = torch.empty(M, max(top_n, 1), dtype=torch.float32, device=dev)
    ti = torch.empty(M, max(top_n, 1), dtype=torch.int32, device=dev)
    ws = torch.empty(max(int(lib().dihip_logprobs_workspace_bytes(M, N, top_n)), 8), dtype=torch.uint8, device=dev)
    check(lib().dihip_logprobs(cur_stream(), ptr(logits), M, N, ptr(chosen), top_n, max(top_n, 1), ptr(tok), ptr(tv), ptr(ti), ptr(ws), ws.numel()),
          "dihip_logprobs")
    return tok, tv[:, :top_n], ti[:, :top_n]
