"""Whole-decoder oracle (TEST INFRASTRUCTURE ONLY): the Qwen2 decode graph of the reference
(python/pyhie/allspark/model/qwen_v15.py:210-388 -- RMSNorm -> qkv GEMM + bias -> Rotary -> attention over the cache ->
o GEMM + residual -> RMSNorm -> SiLU(gate) * up -> down GEMM + residual; final norm -> FT lm_head -> greedy) assembled
from the oracle's pieces: quantised linear (gemm_ref), cache codec (kv_codec), attention and glue.  Rounding points are
the reference's: FT activations between operators, f32 residual stream, f32 logits.

Two evaluation orders of the same function, which must agree (tests/test_oracle_model.py): `step()` decodes token by
token against a growing cache (what the product path does), `last_logits_from_scratch()` recomputes a whole sequence with
the prefill attention oracle.
"""
import numpy as np

from . import attention, gemm_ref, glue, kv_codec
from .numerics import bf16_round


class DecoderOracle:
    def __init__(self, layers, embed, final_norm, lm_head, n_heads, n_kv, head_dim, wbits, group, eps=1e-6,
                 rope_theta=1000000.0, kv_mode="none"):
        """layers: list of dicts with 'qkv', 'o', 'gate', 'up', 'down' = (q, scales, zeros) in the formats of
        gemm_ref.gemm_a16wx, plus 'qkv_bias', 'ln1', 'ln2' (float arrays); embed [V, hidden], lm_head [hidden, V]."""
        self.layers, self.embed, self.final_norm, self.lm_head = layers, embed, final_norm, lm_head
        self.n, self.g, self.H = n_heads, n_kv, head_dim
        self.wbits, self.group, self.eps, self.kv_mode = wbits, group, eps, kv_mode
        self.inv_freq = glue.rope_inv_freq(head_dim, rope_theta)
        self.cache = None

    # -- pieces --------------------------------------------------------------------------------
    def linear(self, x, w, ft, bias=None):
        q, s, z = w
        return gemm_ref.gemm_a16wx(x, q, s, z, self.group, self.wbits, bias=bias, ft=ft)

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
        xn = bf16_round(glue.rmsnorm(h, lw["ln2"], self.eps))
        act = bf16_round(glue.silu(self.linear(xn, lw["gate"], "f32")) * self.linear(xn, lw["up"], "f32"))
        return h + self.linear(act, lw["down"], "f32")

    def _logits(self, h):
        xn = bf16_round(glue.rmsnorm(h, self.final_norm, self.eps))
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
            h = h + self.linear(attn, lw["o"], "f32")
            h = self._mlp(h, lw)
        return self._logits(h)

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
            h = h + self.linear(attn.reshape(L, n * H), lw["o"], "f32")
            h = self._mlp(h, lw)
        return self._logits(h[-1:])
