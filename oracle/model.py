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
tests/test_gpu_decoder.py reports the product's error against both (say which rounding points differ).

Two evaluation orders of the same function, which must agree (tests/test_oracle_model.py): `step()` decodes token by
token against a growing cache (what the product path does), `last_logits_from_scratch()` recomputes a whole sequence with
the prefill attention oracle.
"""
import numpy as np

from . import attention, gemm_ref, glue, kv_codec, moe
from .numerics import bf16_round


class DecoderOracle:
    def __init__(self, layers, embed, final_norm, lm_head, n_heads, n_kv, head_dim, wbits, group, eps=1e-6,
                 rope_theta=1000000.0, kv_mode="none", rounding="x86", cache_weights=False):
        """layers: list of dicts with 'qkv', 'o', 'gate', 'up', 'down' = (q, scales, zeros) in the formats of
        gemm_ref.gemm_a16wx, plus 'qkv_bias', 'ln1', 'ln2' (float arrays); embed [V, hidden], lm_head [hidden, V]."""
        self.layers, self.embed, self.final_norm, self.lm_head = layers, embed, final_norm, lm_head
        self.n, self.g, self.H = n_heads, n_kv, head_dim
        self.wbits, self.group, self.eps, self.kv_mode = wbits, group, eps, kv_mode
        self.inv_freq = glue.rope_inv_freq(head_dim, rope_theta)
        self.cache = None
        assert rounding in ("x86", "ft_graph")
        self.rounding = rounding
        self._wcache = {} if cache_weights else None  # id(q) -> dequantised f64 [K, N] (large models: dequantise once)
        self._lm64 = None

    # -- pieces --------------------------------------------------------------------------------
    def linear(self, x, w, ft, bias=None):
        q, s, z = w
        if self._wcache is None:
            return gemm_ref.gemm_a16wx(x, q, s, z, self.group, self.wbits, bias=bias, ft=ft)
        # same arithmetic as gemm_ref.gemm_a16wx(mode="exact"), with the dequantised matrix kept
        w64 = self._wcache.get(id(q))
        if w64 is None:
            w64 = self._wcache[id(q)] = gemm_ref.dequant(q, s, z, self.group, self.wbits).astype(np.float64)
        v = np.asarray(x, np.float32).astype(np.float64) @ w64
        if bias is not None:
            v = v + np.asarray(bias, np.float64)[None, :]
        from .numerics import ft_round
        return ft_round(v.astype(np.float32), ft)

    def _ft(self, v):
        """An operator output under the bf16 graph of the reference; f32 (no rounding) under x86 semantics."""
        return bf16_round(v) if self.rounding == "ft_graph" else v

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
        q = bf16_round(glue.rope(row[: n * H].reshape(n, H), pos, self.inv_freq))
        k = bf16_round(glue.rope(row[n * H:(n + g) * H].reshape(g, H), pos, self.inv_freq))
        v = row[(n + g) * H:].reshape(g, H)
        return q, k, v

    def _mlp(self, h, lw):
        if "moe" in lw:
            return self._moe_mlp(h, lw)
        xn = bf16_round(glue.rmsnorm(h, lw["ln2"], self.eps))
        gate = self._ft(glue.silu(self.linear(xn, lw["gate"], "f32")))
        up = self._ft(self.linear(xn, lw["up"], "f32"))
        act = bf16_round(gate * up)
        return self._residual(h, self.linear(act, lw["down"], "f32"))

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

    def _logits(self, h):
        xn = bf16_round(glue.rmsnorm(h, self.final_norm, self.eps))
        if self._wcache is not None:
            if self._lm64 is None:
                self._lm64 = self.lm_head.astype(np.float64)
            return (xn.astype(np.float64) @ self._lm64).astype(np.float32)
        return (xn.astype(np.float64) @ self.lm_head.astype(np.float64)).astype(np.float32)

    # -- incremental decode ------------------------------------------------------------------------
    def step(self, ids):
        """One decode step for a batch of independent requests; returns f32 logits [B, V]."""
        n, H = self.n, self.H
        B = len(ids)
        if self.cache is None:
            self.cache = [[([], []) for _ in range(B)] for _ in self.layers]
        h = self.embed[np.asarray(ids)].astype(np.float32)
        for li, lw in enumerate(self.layers):
            xn = bf16_round(glue.rmsnorm(h, lw["ln1"], self.eps))
            qkv = self.linear(xn, lw["qkv"], "bf16", bias=lw["qkv_bias"])
            attn = np.empty((B, n * H), np.float32)
            for b in range(B):
                ks, vs = self.cache[li][b]
                q, k, v = self._qkv_heads(qkv[b], len(ks))
                ks.append(self.kv_store(k))
                vs.append(self.kv_store(v))
                attn[b] = bf16_round(attention.decode_attention(q, np.stack(ks), np.stack(vs), 1.0 / np.sqrt(H))).reshape(-1)
            h = self._residual(h, self.linear(attn, lw["o"], "f32"))
            h = self._mlp(h, lw)
        return self._logits(h)

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
                xn = bf16_round(glue.rmsnorm(h, lw["ln1"], self.eps))
                qkv = self.linear(xn, lw["qkv"], "bf16", bias=lw["qkv_bias"])
                pos = np.arange(L, dtype=np.int32)
                g = self.g
                q = bf16_round(glue.rope(qkv[:, : n * H].reshape(L, n, H), pos, self.inv_freq))
                k = bf16_round(glue.rope(qkv[:, n * H:(n + g) * H].reshape(L, g, H), pos, self.inv_freq))
                v = qkv[:, (n + g) * H:].reshape(L, g, H)
                ks, vs = self.cache[li][b]
                for t in range(L):
                    ks.append(self.kv_store(k[t]))
                    vs.append(self.kv_store(v[t]))
                # the context phase attends over the fresh (unquantised) K / V of the prompt (span_attn_op_cuda.cpp:
                # runContext: xformer_prefill_attention on the qkv rows; the cache copy is a side output)
                attn = bf16_round(attention.prefill_attention(q, k, v, 1.0 / np.sqrt(H), True))
                h = self._residual(h, self.linear(attn.reshape(L, n * H), lw["o"], "f32"))
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
            xn = bf16_round(glue.rmsnorm(h, lw["ln1"], self.eps))
            qkv = self.linear(xn, lw["qkv"], "bf16", bias=lw["qkv_bias"])
            qs, ks, vs = [], [], []
            for t in range(L):
                q, k, v = self._qkv_heads(qkv[t], t)
                qs.append(q)
                ks.append(self.kv_store(k))
                vs.append(self.kv_store(v))
            attn = bf16_round(attention.prefill_attention(np.stack(qs), np.stack(ks), np.stack(vs), 1.0 / np.sqrt(H), True))
            h = self._residual(h, self.linear(attn.reshape(L, n * H), lw["o"], "f32"))
            h = self._mlp(h, lw)
        return self._logits(h[-1:])
