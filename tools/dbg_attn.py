# debug: error statistics of the u4 MFMA decode attention vs the oracle
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest  # noqa
import test_gpu_kv_attn as T
import importlib.util
def load_pkg():
    import importlib, types
    root = os.path.join(os.path.dirname(__file__), "..")
    spec = importlib.util.spec_from_file_location("dash_infer_amd", os.path.join(root, "dash-infer_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(root, "dash-infer_amd")])
    m = importlib.util.module_from_spec(spec); sys.modules["dash_infer_amd"] = m; spec.loader.exec_module(m)
load_pkg()
from dash_infer_amd import ops
for (n, g, S, lens) in [(14, 2, 16, [999]), (28, 4, 128, [5, 700, 130]), (8, 1, 32, [257, 64])]:
    rng = np.random.default_rng(5 + S)
    H, ft = 128, "bf16"
    pool, kv, ok, ov = T.build_batch(ops, rng, lens, n, g, H, S, "u4", ft)
    q = T.bf16_round(rng.normal(0, 1, (len(lens), n, H)).astype(np.float32))
    scale = 1.0 / np.sqrt(H)
    out = T.run_attn(ops, kv, q, lens, n, g, H, ft, scale)
    ref = T.oracle_attn(ok, ov, q, lens, scale)
    for b, L in enumerate(lens):
        e = np.abs(out[b] - ref[b])
        bad = e > (2.5e-3 + 1e-2 * np.abs(ref[b]))
        print(n, g, S, "len", L, "max abs err %.5f" % e.max(), "mean %.6f" % e.mean(), "bad", int(bad.sum()), "of", bad.size,
              "max|ref| %.3f" % np.abs(ref[b]).max(), "bad heads", sorted(set(np.argwhere(bad)[:, 0].tolist()))[:10])
