"""Whole-decoder oracle (TEST INFRASTRUCTURE ONLY): the Qwen2 decode graph of the reference
(python/pyhie/allspark/model/qwen_v15.py:210-388 -- RMSNorm -> qkv GEMM + bias -> Rotary -> attention over the cache ->
o GEMM + residual -> RMSNorm -> SiLU(gate) * up -> down GEMM + residual; final norm -> FT lm_head -> greedy) assembled
from the oracle's pieces: quantised linear (gemm_ref), cache codec (kv_codec), attention and glue.

Two sets of rounding points (`rounding=`):
  "x86"      (default; the north star's comparison target, "the reference x86 CPU path"): activations between operators are
             f32 on x86 (fp32 activations, bf16 src + weights inside the matmul under medium_bf16, f32 out:
             gemm_op_cpu.cpp:75-126) -- so the residual stream is f32, SiLU(gate) and up meet as f32 (oneDNN post-ops) and
             the product is rounded once, where the down matmul takes its bf16 src.  Not x86: qkv, the cache and the
             attention output are rounded to the GPU cache's FT (bf16), because the product's KV cache is FT.
  "ft_graph" the CUDA bf16 graph of the reference (qwen_v15.py:296-346 with FT = bf16): every operator output is an FT
             tensor -- the residual ADD results, Gemm(gate)+SiLU, Gemm(up), their MUL, the o / down GEMM outputs.
  "x86_pure_bf16" the x86 path itself under matmul_precision=medium_bf16 (gemm_op_cpu.cpp:75-126, gemm_op_x86_spr.cpp:55-69):
             every tensor between operators is f32 -- qkv, the rotated q / k, the KV cache (BatchMQAOp's contiguous f32
             cache, batch_mqa_op.cpp:140-179) and the attention output (cblas_sgemm + f32 softmax) included; inside a matmul
             the src is converted to bf16 and the weight is the bf16 reorder of the dequantised (q - z) * s (x86 has
             no weight-only GEMM, the quantised configs run on dequantised weights), f32 accumulation and output.
  "x86_pure_f32"  the same path under the default matmul precision: f32 src and f32 dequantised weights, nothing rounded.
  "x86_pure_bf16_exactw"  ablation: medium_bf16's src conversion but the unrounded (q - z) * s -- what separates the product's
             A16Wx arithmetic (bf16 activations into the GEMM, exact integer weights x f32 scale) from "x86_pure_bf16" is the
             weight rounding, from "x86" the FT rounding of qkv / cache / attention output.
tests/test_gpu_decoder.py reports the product's error against these (say which rounding points differ).

Two evaluation orders of the same function, which must agree (tests/test_oracle_model.py): `step()` decodes token by
token against a growing cache (what the product path does), `last_logits_from_scratch()` recomputes a whole sequence with
the prefill attention oracle.
"""
import numpy as np

from . import attention, gemm_ref, glue, kv_codec, moe
from .numerics import bf16_round, ft_round as _ft_round


class Rounding:
    """Where one evaluation of the decoder graph rounds (see the module docstring)."""

    def __init__(self, name, op_out_ft, gemm_src_ft, w_bf16, qkv_ft, attn_ft):
        self.name = name
        self.op_out_ft = op_out_ft      # every operator output is an FT tensor (the CUDA bf16 graph)
        self.gemm_src_ft = gemm_src_ft  # a matmul takes its src as bf16 (A16Wx kernels; oneDNN under medium_bf16)
        self.w_bf16 = w_bf16            # the matmul's weight is the bf16 rounding of the dequantised value
        self.qkv_ft = qkv_ft            # qkv GEMM output, rotated q / k and the cache rows are FT
        self.attn_ft = attn_ft          # the attention output is FT

    def __eq__(self, other):            # `oracle.rounding == "ft_graph"` keeps working
        return self.name == (other.name if isinstance(other, Rounding) else other)

    def __repr__(self):
        return f"Rounding({self.name})"


ROUNDINGS = {r.name: r for r in (
    Rounding("x86", False, True, False, True, True),
    Rounding("ft_graph", True, True, False, True, True),
    Rounding("x86_pure_bf16", False, True, True, False, False),
    Rounding("x86_pure_bf16_exactw", False, True, False, False, False),
    Rounding("x86_pure_f32", False, False, False, False, False),
)}


class DecoderOracle:
    def __init__(self, layers, embed, final_norm, lm_head, n_heads, n_kv, head_dim, wbits, group, eps=1e-6,
                 rope_theta=1000000.0, kv_mode="none", rounding="x86", cache_weights=False, ft="bf16"):
        """layers: list of dicts with 'qkv', 'o', 'gate', 'up', 'down' = (q, scales, zeros) in the formats of
        gemm_ref.gemm_a16wx -- or a plain FT-valued float array [K, N] for an UNQUANTISED layer (op type Gemm: BASELINE
        configs[0], Qwen2-0.5B bf16 on the x86 path; wbits = 16) --, plus 'qkv_bias', 'ln1', 'ln2' (float arrays);
        embed [V, hidden], lm_head [hidden, V].  Any head size (Qwen2-0.5B: 64)."""
        self.layers, self.embed, self.final_norm, self.lm_head = layers, embed, final_norm, lm_head
        self.n, self.g, self.H = n_heads, n_kv, head_dim
        self.wbits, self.group, self.eps, self.kv_mode = wbits, group, eps, kv_mode
        self.ft = ft                    # the activation type "FT" of the graph: bf16 (default) or f16
        self.inv_freq = glue.rope_inv_freq(head_dim, rope_theta)
        self.cache = None
        self.rounding = rounding
        self.acc = np.float64           # accumulation type of the matmuls / attention (np.float32: the full-depth comparisons)
        self.threads = 1                # thread pool width of the per-head attention loop (full-depth comparisons)
        self._wcache = {} if cache_weights else None  # id(q) -> dequantised f64 [K, N] (large models: dequantise once)
        self._lm64 = None

    @property
    def rounding(self):
        return self._rounding

    @rounding.setter
    def rounding(self, r):
        self._rounding = r if isinstance(r, Rounding) else ROUNDINGS[r]

    def _r(self, x, threads=1):
        """round to FT (the 16-bit activation type of this model)"""
        return bf16_round(x, threads=threads) if self.ft == "bf16" else _ft_round(x, self.ft)

    # -- pieces --------------------------------------------------------------------------------
    def dequantised(self, w):
        """The matmul's weight matrix [K, N] in the accumulation type: (q - z) * s in f32 (bit-exact restatement of the
        reference's host loop), rounded to bf16 where the x86 path holds bf16 weights."""
        if isinstance(w, np.ndarray):   # unquantised FT weight (already bf16-valued: the bf16 reorder is the identity)
            return np.asarray(w, np.float32).astype(self.acc, copy=False)
        q, s, z = w
        w32 = gemm_ref.dequant(q, s, z, self.group, self.wbits)
        if self.rounding.w_bf16:
            w32 = bf16_round(w32)
        return w32.astype(self.acc, copy=False)

    def linear(self, x, w, ft, bias=None, W=None):
        """x . W (+ bias), rounded to `ft`.  W: the matrix already dequantised for this rounding (lockstep evaluation of
        several oracles over one dequantisation, `teacher_forced_logits`), else taken from / put into the weight cache."""
        q = w if isinstance(w, np.ndarray) else w[0]
        if W is None and self._wcache is not None:
            W = self._wcache.get((id(q), self.rounding.w_bf16))
            if W is None:
                W = self._wcache[(id(q), self.rounding.w_bf16)] = self.dequantised(w)
        if W is None:
            W = self.dequantised(w)
        v = np.asarray(x, np.float32).astype(self.acc, copy=False) @ W
        if bias is not None:
            v = v + np.asarray(bias, self.acc)[None, :]
        from .numerics import ft_round
        return ft_round(v.astype(np.float32, copy=False), ft)

    def _src(self, x):
        """What a matmul reads: bf16 under medium_bf16 / the A16Wx kernels, the f32 tensor itself otherwise."""
        return self._r(x) if self.rounding.gemm_src_ft else np.asarray(x, np.float32)

    def _qkv_ft(self):
        return self.ft if self.rounding.qkv_ft else "f32"

    def _attn_out(self, v):
        return self._r(v) if self.rounding.attn_ft else np.asarray(v, np.float32)

    def _ft(self, v):
        """An operator output under the bf16 graph of the reference; f32 (no rounding) under x86 semantics."""
        return self._r(v) if self.rounding.op_out_ft else v

    def _residual(self, h, y):
        return self._ft(self._ft(h) + self._ft(y))

    def kv_store(self, x):
        """What the cache returns for rows x [g, H] written at this step."""
        if self.kv_mode == "none":
            return x
        zero, scale = kv_codec.quant_params(x, self.kv_mode)
        return kv_codec.dequantize(kv_codec.quantize(x, zero, scale, self.kv_mode), zero, scale)

    def _qkv_heads(self, row, pos):
        n, g, H = self.n, self.g, self.H
        rq = self._r if self.rounding.qkv_ft else (lambda t: t)
        q = rq(glue.rope(row[: n * H].reshape(n, H), pos, self.inv_freq))
        k = rq(glue.rope(row[n * H:(n + g) * H].reshape(g, H), pos, self.inv_freq))
        v = row[(n + g) * H:].reshape(g, H)
        return q, k, v

    def _mlp(self, h, lw, W=None):
        if "moe" in lw:
            return self._moe_mlp(h, lw)
        W = W or {}
        xn = self._src(glue.rmsnorm(h, lw["ln2"], self.eps))
        gate = self._ft(glue.silu(self.linear(xn, lw["gate"], "f32", W=W.get("gate")), dtype=self.acc))
        up = self._ft(self.linear(xn, lw["up"], "f32", W=W.get("up")))
        act = self._src(gate * up)
        return self._residual(h, self.linear(act, lw["down"], "f32", W=W.get("down")))

    def _moe_mlp(self, h, lw):
        """The mixture-of-experts layer of python/pyhie/allspark/model/qwen_v20_moe.py:318-382: router Gemm -> MOE (softmax,
        top-k, expert FFNs, combine: oracle/moe.py) ; shared expert (gate_up Gemm -> UnaryGLU -> down Gemm) scaled by its
        sigmoid gate (Gemm with activation SIGMOID, N = 1; CalcExpert, calc_expert.cu:27-35) ; expert_add ; final_add.
        lw["moe"]: router f32 [hidden, E], top_k, experts_gate / _up / _down: lists of (q, s, z), shared_gate_w f32 [hidden, 1];
        the shared expert's projections are lw["gate"] / ["up"] / ["down"]."""
        m = lw["moe"]
        xn = bf16_round(glue.rmsnorm(h, lw["ln2"], self.eps))
        x64 = xn.astype(np.float64)
        logits = bf16_round((x64 @ np.asarray(m["router"], np.float64)).astype(np.float32))          # Gemm output, FT
        scores, experts = moe.route(logits, m["top_k"])
        moe_out = moe.experts_ffn(xn, experts, scores, m["experts_gate"], m["experts_up"], m["experts_down"], self.group, self.wbits, ft="bf16")
        gate = self._ft(glue.silu(self.linear(xn, lw["gate"], "f32")))
        up = self._ft(self.linear(xn, lw["up"], "f32"))
        act = bf16_round(gate * up)
        shared = bf16_round(self.linear(act, lw["down"], "f32"))                                       # Gemm output, FT
        z = (x64 @ np.asarray(m["shared_gate_w"], np.float64)).astype(np.float32)                    # [B, 1]
        sig = bf16_round((1.0 / (1.0 + np.exp(-z.astype(np.float64)))).astype(np.float32))
        calc = bf16_round(shared * sig)                                                                # CalcExpert output, FT
        if self.rounding == "ft_graph":
            return bf16_round(bf16_round(moe_out + calc) + bf16_round(h))                              # expert_add, final_add
        return ((h + moe_out) + calc).astype(np.float32)

    def _logits(self, h, lm=None):
        xn = self._src(glue.rmsnorm(h, self.final_norm, self.eps))
        if lm is None:
            if self._wcache is not None:
                if self._lm64 is None or self._lm64.dtype != self.acc:
                    self._lm64 = self.lm_head.astype(self.acc)
                lm = self._lm64
            else:
                lm = self.lm_head.astype(self.acc)
        return (xn.astype(self.acc, copy=False) @ lm).astype(np.float32)

    # -- incremental decode ------------------------------------------------------------------------
    def step(self, ids):
        """One decode step for a batch of independent requests; returns f32 logits [B, V]."""
        n, H = self.n, self.H
        B = len(ids)
        if self.cache is None:
            self.cache = [[([], []) for _ in range(B)] for _ in self.layers]
        h = self.embed[np.asarray(ids)].astype(np.float32)
        for li, lw in enumerate(self.layers):
            xn = self._src(glue.rmsnorm(h, lw["ln1"], self.eps))
            qkv = self.linear(xn, lw["qkv"], self._qkv_ft(), bias=lw["qkv_bias"])
            attn = np.empty((B, n * H), np.float32)
            for b in range(B):
                ks, vs = self.cache[li][b]
                q, k, v = self._qkv_heads(qkv[b], len(ks))
                ks.append(self.kv_store(k))
                vs.append(self.kv_store(v))
                attn[b] = self._attn_out(attention.decode_attention(q, np.stack(ks), np.stack(vs), 1.0 / np.sqrt(H))).reshape(-1)
            h = self._residual(h, self.linear(self._src(attn), lw["o"], "f32"))
            h = self._mlp(h, lw)
        return self._logits(h)

    def _context_qkv(self, h, lw, L, W=None):
        """q [L, n, H], k, v [L, g, H] of a whole prompt (positions 0 .. L-1) as the attention sees them."""
        n, g, H = self.n, self.g, self.H
        xn = self._src(glue.rmsnorm(h, lw["ln1"], self.eps))
        qkv = self.linear(xn, lw["qkv"], self._qkv_ft(), bias=lw["qkv_bias"], W=(W or {}).get("qkv"))
        pos = np.arange(L, dtype=np.int32)
        rq = self._r if self.rounding.qkv_ft else (lambda t: t)
        q = rq(glue.rope(qkv[:, : n * H].reshape(L, n, H), pos, self.inv_freq))
        k = rq(glue.rope(qkv[:, n * H:(n + g) * H].reshape(L, g, H), pos, self.inv_freq))
        v = qkv[:, (n + g) * H:].reshape(L, g, H)
        return q, k, v

    # -- context phase: fills the cache of every request, returns the logits after each prompt's last token --------
    def prefill(self, seqs):
        """seqs: one token-id list per request.  Same function as feeding the tokens one by one through step(); the
        positions of a prompt are computed together with the causal prefill attention (what the reference's context
        phase does).  Leaves self.cache ready for step()."""
        n, H = self.n, self.H
        self.cache = [[([], []) for _ in seqs] for _ in self.layers]
        out = []
        for b, seq in enumerate(seqs):
            L = len(seq)
            h = self.embed[np.asarray(seq)].astype(np.float32)
            for li, lw in enumerate(self.layers):
                q, k, v = self._context_qkv(h, lw, L)
                ks, vs = self.cache[li][b]
                for t in range(L):
                    ks.append(self.kv_store(k[t]))
                    vs.append(self.kv_store(v[t]))
                # the context phase attends over the fresh (unquantised) K / V of the prompt (span_attn_op_cuda.cpp:
                # runContext: xformer_prefill_attention on the qkv rows; the cache copy is a side output)
                attn = self._attn_out(attention.prefill_attention(q, k, v, 1.0 / np.sqrt(H), True, dtype=self.acc, threads=self.threads))
                h = self._residual(h, self.linear(self._src(attn.reshape(L, n * H)), lw["o"], "f32"))
                h = self._mlp(h, lw)
            out.append(self._logits(h[-1:])[0])
        return np.stack(out)

    # -- the same function, evaluated over a whole sequence at once ---------------------------------------
    def last_logits_from_scratch(self, seq):
        """Logits after the last token of ONE sequence, all positions computed together with causal prefill attention."""
        n, H = self.n, self.H
        L = len(seq)
        h = self.embed[np.asarray(seq)].astype(np.float32)
        for lw in self.layers:
            xn = self._src(glue.rmsnorm(h, lw["ln1"], self.eps))
            qkv = self.linear(xn, lw["qkv"], self._qkv_ft(), bias=lw["qkv_bias"])
            qs, ks, vs = [], [], []
            for t in range(L):
                q, k, v = self._qkv_heads(qkv[t], t)
                qs.append(q)
                ks.append(self.kv_store(k))
                vs.append(self.kv_store(v))
            attn = self._attn_out(attention.prefill_attention(np.stack(qs), np.stack(ks), np.stack(vs), 1.0 / np.sqrt(H), True))
            h = self._residual(h, self.linear(self._src(attn.reshape(L, n * H)), lw["o"], "f32"))
            h = self._mlp(h, lw)
        return self._logits(h[-1:])

    def scratch_layer(self, h, lw, W=None):
        """One decoder layer over ALL positions of one sequence (h [L, hidden]); 16-bit / f32 cache only (the cache returns
        what was written).  W: {"qkv" | "o" | "gate" | "up" | "down": matrix dequantised for this rounding} or None."""
        assert self.kv_mode == "none"
        L = h.shape[0]
        q, k, v = self._context_qkv(h, lw, L, W)
        attn = self._attn_out(attention.prefill_attention(q, k, v, 1.0 / np.sqrt(self.H), True, dtype=self.acc, threads=self.threads))
        h = self._residual(h, self.linear(self._src(attn.reshape(L, self.n * self.H)), lw["o"], "f32", W=(W or {}).get("o")))
        return self._mlp(h, lw, W)


def teacher_forced_logits(oracles, layers, seq, n_last, progress=None, threads=1):
    """Logits [n_last, V] of the last n_last positions of `seq` for each oracle (same weights, different Rounding), all
    positions computed together with the causal attention oracle -- for a greedy run that is fed the tokens the product
    chose, this is the same function as prefill + step() token by token (tests/test_oracle_model.py) at one pass over
    the weights: every layer is dequantised ONCE (twice when some oracle wants the bf16 weight rounding of the x86 path)
    and applied to every oracle's hidden rows before the next layer is touched, so a 28-layer model at Qwen2-7B widths
    needs one layer's matrices in host memory at a time.  `layers`: a sequence of layer dicts (may build each on access)."""
    hs = [o.embed[np.asarray(seq)].astype(np.float32) for o in oracles]
    o0 = oracles[0]
    for o in oracles:
        o.threads = max(o.threads, threads)
    for li in range(len(layers)):
        lw = layers[li]
        mats = {}
        for name in ("qkv", "o", "gate", "up", "down"):
            w32 = (np.asarray(lw[name], np.float32) if isinstance(lw[name], np.ndarray)
                   else gemm_ref.dequant(*lw[name], o0.group, o0.wbits, threads=threads))
            mats[name] = {False: w32}
            if any(o.rounding.w_bf16 for o in oracles):
                mats[name][True] = bf16_round(w32, threads=threads)
        for i, o in enumerate(oracles):
            W = {name: mats[name][o.rounding.w_bf16].astype(o.acc, copy=False) for name in mats}
            hs[i] = o.scratch_layer(hs[i], lw, W)
        del mats, lw
        if progress:
            progress(li)
    out = []
    for o, h in zip(oracles, hs):
        lm = o.lm_head.astype(o.acc, copy=False)
        out.append(o._logits(h[-n_last:], lm=lm))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Layer-major evaluation of whole teacher-forced runs (full depth, any cache mode, a batch of ragged requests).
def _layer_sequences(o, hs, lw, W, prompt_lens, want_kv=False):
    """One decoder layer over ALL positions of several independent sequences under oracle `o`'s rounding and cache mode.
    hs[b]: f32 [L_b, hidden].  Rows below prompt_lens[b] are CONTEXT-phase rows: causal attention over the fresh (unquantised)
    K / V of the prompt (SpanAttnOp::runContext attends over the qkv rows; the cache copy is a side output).  Rows from
    prompt_lens[b] on are DECODER-phase rows: row t attends over what the cache returns for rows 0 .. t (kv_store: the codec
    image for the int8 / uint4 caches) -- exactly step() after prefill(), evaluated layer by layer.
    -> (list of f32 [L_b, hidden] outputs; with want_kv a list of (k, v, k_image, v_image), each [L_b, g, H]: the FT-valued rows
    written to the cache and what the cache returns for them -- the same arrays for the 16-bit cache)."""
    n, g, H = o.n, o.g, o.H
    W = W or {}
    rows = np.concatenate(hs)
    xn = o._src(glue.rmsnorm(rows, lw["ln1"], o.eps))
    qkv = o.linear(xn, lw["qkv"], o._qkv_ft(), bias=lw["qkv_bias"], W=W.get("qkv"))
    rq = o._r if o.rounding.qkv_ft else (lambda t: t)
    alpha = 1.0 / np.sqrt(H)
    attn_rows, kvs, off = [], [], 0
    for b, h in enumerate(hs):
        L, P = h.shape[0], int(prompt_lens[b])
        r = qkv[off:off + L]
        off += L
        pos = np.arange(L, dtype=np.int32)
        q = rq(glue.rope(r[:, : n * H].reshape(L, n, H), pos, o.inv_freq))
        k = rq(glue.rope(r[:, n * H:(n + g) * H].reshape(L, g, H), pos, o.inv_freq))
        v = r[:, (n + g) * H:].reshape(L, g, H)
        out = np.empty((L, n, H), np.float32)
        if P > 0:
            out[:P] = attention.prefill_attention(q[:P], k[:P], v[:P], alpha, True, dtype=o.acc, threads=o.threads)
        if o.kv_mode == "none":
            kc, vc = k, v
        else:
            zk, sk = kv_codec.quant_params(k, o.kv_mode)
            zv, sv = kv_codec.quant_params(v, o.kv_mode)
            kc = kv_codec.dequantize(kv_codec.quantize(k, zk, sk, o.kv_mode), zk, sk)
            vc = kv_codec.dequantize(kv_codec.quantize(v, zv, sv, o.kv_mode), zv, sv)
        if L > P:
            # decoder row t sees the cache's image of rows 0 .. t: the last L - P rows of one causal pass over it
            out[P:] = attention.prefill_attention(q[P:], kc, vc, alpha, True, dtype=o.acc, threads=o.threads)
        attn_rows.append(o._attn_out(out).reshape(L, n * H))
        if want_kv:
            kvs.append((k, v, kc, vc))
    h2 = o._residual(rows, o.linear(o._src(np.concatenate(attn_rows)), lw["o"], "f32", W=W.get("o")))
    h3 = o._mlp(h2, lw, W)
    outs, off = [], 0
    for h in hs:
        outs.append(h3[off:off + h.shape[0]])
        off += h.shape[0]
    return outs, kvs


def teacher_forced_trace(o, layers, seqs, prompt_lens, on_layer=None, threads=1, want_kv=False, on_logits=None):
    """Teacher-forced evaluation of a batch of requests at full depth, one pass over the weights (each layer dequantised once
    and applied to every row of every request before the next layer is touched).  seqs[b] = prompt tokens followed by the
    tokens fed at the decode steps; prompt_lens[b] = how many of them were the context phase.  Any cache mode.
    on_layer(li, h_in, h_out, kvs): per layer, the lists of f32 [L_b, hidden] layer inputs / outputs and (with want_kv) each
    request's (k, v, k_image, v_image) -- the per-layer drift test feeds these to the GPU layer.
    -> list of f32 logits [L_b - prompt_lens[b] + 1, V]: after the prompt's last token, then after every decode step;
    with on_logits(b, logits_b) each request's block is handed over and dropped instead (batch 32 x 65 rows x 152k columns)."""
    o.threads = max(o.threads, threads)
    hs = [o.embed[np.asarray(s)].astype(np.float32) for s in seqs]
    for li in range(len(layers)):
        lw = layers[li]
        mats = {}
        for name in ("qkv", "o", "gate", "up", "down"):
            w32 = (np.asarray(lw[name], np.float32) if isinstance(lw[name], np.ndarray)
                   else gemm_ref.dequant(*lw[name], o.group, o.wbits, threads=threads))
            if o.rounding.w_bf16:
                w32 = bf16_round(w32, threads=threads)
            mats[name] = w32.astype(o.acc, copy=False)
        outs, kvs = _layer_sequences(o, hs, lw, mats, prompt_lens, want_kv=want_kv)
        if on_layer is not None:
            on_layer(li, hs, outs, kvs)
        hs = outs
        del mats, lw
    lm = o.lm_head.astype(o.acc, copy=False)
    if on_logits is not None:
        for b, (h, P) in enumerate(zip(hs, prompt_lens)):
            on_logits(b, o._logits(h[int(P) - 1:], lm=lm))
        return None
    return [o._logits(h[int(P) - 1:], lm=lm) for h, P in zip(hs, prompt_lens)]
