#!/bin/bash
# round 5, call w: context-phase GEMM with the tail of its grid split in K: tests, then the prefill workload with it on / off
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5w
{
timeout 1200 python -m pytest tests/test_gpu_gemm.py -q -m gpu --timeout 900 -s -k "prefill" 2>&1 | grep -E "prefill tail split|passed|failed|Error|assert" | cut -c1-300 | tail -20
for d in 1 0; do
  DIHIP_PREFILL_TAIL_SPLIT=$d timeout 300 python bench.py --workload prefill_2048 --no-cpu-baseline --no-extra > gpurun_out/r5w/bench_prefill_split$d.json 2> gpurun_out/r5w/bench_prefill_split$d.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r5w/bench_prefill_split$d.json"))
    print("prefill_2048 split=$d", d["value"], d["ms_per_step"], {k: (v["avg_us"], v["tflops"]) for k, v in d.get("gemms", {}).items()}, d.get("attention", {}).get("avg_launch_us"))
except Exception as e:
    print("prefill split=$d FAILED", e)
PY
done
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_host_runner.py -q -m gpu --timeout 600 -k "prefill or context or prefix" 2>&1 | tail -3
} 2>&1 | tee gpurun_out/r5w/log.txt
