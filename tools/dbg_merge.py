import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import decoder, ops
os.environ["DIHIP_DECODER_ATTN_MERGE"] = "1"
cfg = decoder.QWEN2_7B
model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128, True), layers=2)
sess = decoder.DecodeSession(model, 1, 2048 + 64, span_len=128, kv_mode="none")
sess.fill_cache_random(2048)
sess.set_state([5], [2048])
print("nsplits", sess.attn_nsplits, flush=True)
m, sc = model, sess.scratch
lw = m.layers[0]
ops.embedding(sess.ids, m.embed, out=sess.h); torch.cuda.synchronize(); print("emb ok", flush=True)
ops.fused_norm_gemm(sess.h, lw.ln1, cfg.eps, lw.qkv, lw.qkv_bias, sc, out=sess.qkv); torch.cuda.synchronize(); print("qkv ok", flush=True)
ops.span_attn_decode_fused_partials(sess.qkv, sess.kv[0], sess.old_lens, sess.rope_tab, sess.n_loc, sess.g_loc, sess.H, sess.max_len, sess.scale, sess.attn_partials)
torch.cuda.synchronize(); print("attn partials ok", flush=True)
ops.fused_attnmerge_gemm_addto(sess.attn_partials, sess.attn_nsplits, sess.n_loc, lw.o, sess.h, sc, out=sess.h, M=1)
torch.cuda.synchronize(); print("o merge ok", flush=True)
for i in range(3):
    sess.step(); torch.cuda.synchronize(); print("step ok", i, sess.ids.tolist(), flush=True)
sess.capture(warmup=1); torch.cuda.synchronize(); print("capture ok", flush=True)
for i in range(3):
    sess.replay(); torch.cuda.synchronize(); print("replay ok", i, flush=True)
