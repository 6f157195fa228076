#!/bin/bash
# round 5, call aa: the int8 cache's decode step in one launch (span_attn_ft_mfma_kernel<FT, I8, FUSED>): attention tests, decoder / host-runner
# tests with the int8 cache, per-layer timing of the step against the two launches (DIHIP_ATTN_I8_FUSED=0)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5aa
{
timeout 1500 python -m pytest tests/test_gpu_kv_attn.py -q -m gpu --timeout 900 -x -k "fused_rope or decode_step or frag32 or long_context or quantised" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_host_runner.py -q -m gpu --timeout 900 -k "i8" 2>&1 | tail -4
python - <<'PY'
import os, sys, torch, time
sys.path.insert(0, os.getcwd())
from tests.conftest import load_pkg
load_pkg()
from dash_infer_amd import decoder, ops
for batch in (1, 32):
    for fused in ("1", "0"):
        os.environ["DIHIP_ATTN_I8_FUSED"] = fused
        cfg = decoder.ModelConfig("b", hidden=3584, layers=4, n_heads=28, n_kv=4, head_dim=128, inter=18944, vocab=1024)
        model = decoder.build_random_model(cfg, decoder.QuantSpec(4, 128), seed=1)
        s = decoder.DecodeSession(model, batch, max_len=2048 + 64, span_len=128, kv_mode="i8")
        s.fill_cache_random(2040, seed=3)
        s.set_state([1] * batch, [2040] * batch)
        s.capture()
        for _ in range(3):
            s.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            s.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print(f"int8 cache, batch {batch}, 4 layers, python session step_attention={s.step_attention} (env only read by the session here): {dt * 1e6 / 4:.2f} us per layer incl. 1/4 of the head")
PY
} 2>&1 | tee gpurun_out/r5aa/log.txt
