import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _load_pkg
_load_pkg()
from dash_infer_amd import decoder, ops
os.environ["DIHIP_DECODER_ATTN_MERGE"] = "1"
cfg = decoder.QWEN2_7B
model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128, True), layers=2)
sess = decoder.DecodeSession(model, 1, 2048 + 88, span_len=128, kv_mode="none")
sess.fill_cache_random(2048)
sess.set_state([5], [2048])
# wrap the ops with syncs
for name in ("fused_norm_gemm", "span_attn_decode_fused_partials", "fused_attnmerge_gemm_addto", "fused_norm_swiglu", "fused_gemm_addto", "lm_head", "argmax", "embedding"):
    f = getattr(ops, name)
    def mk(f, name):
        def g(*a, **k):
            r = f(*a, **k)
            torch.cuda.synchronize()
            print("   ", name, "ok", flush=True) if os.environ.get("VERBOSE") else None
            return r
        return g
    setattr(ops, name, mk(f, name))
for i in range(80):
    if i >= int(os.environ.get("VSTART", "1000")):
        os.environ["VERBOSE"] = "1"
    sess.step()
    print("step", i, "len", int(sess.old_lens[0]), flush=True)
