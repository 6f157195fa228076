#!/bin/bash
# round 5, call q: deferred RMSNorm (o-projection -> gate/up without a norm launch): kernel tests, the batched decoder / host-runner
# tests, then the two small-batch workloads with it on / off (DIHIP_DEFER_RMSNORM=0)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5q
{
timeout 900 python -m pytest tests/test_gpu_deferred_norm.py -q -m gpu -x --timeout 600 -s 2>&1 | grep -E "deferred norm|passed|failed|Error|error|assert" | cut -c1-300 | tail -30
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu --timeout 600 -k "norm or small_batch or kslice or panel or frag32" 2>&1 | tail -4
for w in int4_b32_u4kv cfg3_rank; do
  for d in 1 0; do
    DIHIP_DEFER_RMSNORM=$d timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra > gpurun_out/r5q/bench_${w}_defer$d.json 2> gpurun_out/r5q/bench_${w}_defer$d.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r5q/bench_${w}_defer$d.json"))
    print("$w defer=$d", d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d.get("kernels", {}).items()}, (d.get("host_runner") or {}).get("fused_graph", {}).get("tokens_per_s"))
except Exception as e:
    print("$w defer=$d FAILED", e)
PY
  done
done
} 2>&1 | tee gpurun_out/r5q/log.txt
