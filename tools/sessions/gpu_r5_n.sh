#!/bin/bash
# round 5, call n: distributed merge with the unrolled 4-lanes-per-element merge and no sentinel wait: parity, per-layer time
# (against the sentinel kept, and the last-arriver merge), timeline, headline
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5n
{
timeout 600 python -m pytest tests/test_gpu_attn_block.py -q -m gpu -x --timeout 300 2>&1 | tail -5
echo "== per layer, distributed merge"
timeout 300 python tools/attn_block_trace.py 2>&1 | grep -v amdgpu.ids
echo "== per layer, distributed merge + sentinel wait"
DIHIP_ATTN_BLOCK_SENTINEL=1 timeout 300 python tools/attn_block_trace.py 2>&1 | grep -v amdgpu.ids
echo "== per layer, last-arriver merge"
DIHIP_ATTN_BLOCK_DIST_MERGE=0 timeout 300 python tools/attn_block_trace.py 2>&1 | grep -v amdgpu.ids
echo "== timeline (trace build), distributed merge"
DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/attn_block_trace.py 2>&1 | grep -v amdgpu.ids
echo "== headline"
timeout 600 python bench.py --no-cpu-baseline --no-extra > gpurun_out/r5n/bench_dist.json 2> gpurun_out/r5n/bench_dist.err
DIHIP_ATTN_BLOCK_DIST_MERGE=0 timeout 600 python bench.py --no-cpu-baseline --no-extra > gpurun_out/r5n/bench_last.json 2> gpurun_out/r5n/bench_last.err
python - <<'PY'
import json
for n in ("dist", "last"):
    try:
        d = json.load(open(f"gpurun_out/r5n/bench_{n}.json"))
        print(n, d["value"], d["ms_per_step"], "python runner", d["python_runner"]["tokens_per_s"], {k: v["avg_us"] for k, v in d["kernels"].items()})
    except Exception as e:
        print(n, "FAILED", e)
PY
} 2>&1 | tee gpurun_out/r5n/log.txt
