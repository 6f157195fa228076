#!/bin/bash
# round 5, call u: deferred RMSNorm with the K-slice consumer as its own instantiation (combine in the prologue): tests, stand-alone
# timing, the two small-batch workloads on / off
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5u
{
timeout 900 python -m pytest tests/test_gpu_deferred_norm.py -q -m gpu --timeout 600 -s > gpurun_out/r5u/pytest.log 2>&1
grep -E "^\[deferred norm" gpurun_out/r5u/pytest.log | cut -c1-330
grep -E "passed|failed|Error" gpurun_out/r5u/pytest.log | tail -5
for m in 32 16; do timeout 300 python tools/defer_norm_bench.py $m 2>&1 | grep -v amdgpu.ids; done
for w in int4_b32_u4kv cfg3_rank; do
  for d in 1 0; do
    DIHIP_DEFER_RMSNORM=$d timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra > gpurun_out/r5u/bench_${w}_defer$d.json 2> gpurun_out/r5u/bench_${w}_defer$d.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r5u/bench_${w}_defer$d.json"))
    print("$w defer=$d", d["value"], d["ms_per_step"], (d.get("host_runner") or {}).get("fused_graph", {}).get("tokens_per_s"), d.get("python_runner"))
except Exception as e:
    print("$w defer=$d FAILED", e)
PY
  done
done
} 2>&1 | tee gpurun_out/r5u/log.txt
