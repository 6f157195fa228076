#!/bin/bash
# round 5, call r: deferred RMSNorm after the consumers' partial sums were shortened: tests, A/B of the two small-batch workloads,
# rocprofv3 kernel statistics of the batch-32 workload with it on / off
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5r
{
timeout 900 python -m pytest tests/test_gpu_deferred_norm.py -q -m gpu --timeout 600 -s > gpurun_out/r5r/pytest.log 2>&1
grep -E "^\[deferred norm" gpurun_out/r5r/pytest.log | cut -c1-500
grep -E "passed|failed|Error" gpurun_out/r5r/pytest.log | tail -12
for w in int4_b32_u4kv cfg3_rank; do
  for d in 1 0; do
    DIHIP_DEFER_RMSNORM=$d timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra > gpurun_out/r5r/bench_${w}_defer$d.json 2> gpurun_out/r5r/bench_${w}_defer$d.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r5r/bench_${w}_defer$d.json"))
    print("$w defer=$d", d["value"], d["ms_per_step"], (d.get("host_runner") or {}).get("fused_graph", {}).get("tokens_per_s"), d.get("python_runner"))
except Exception as e:
    print("$w defer=$d FAILED", e)
PY
  done
done
export TMPDIR=/tmp
cd /tmp
for w in int4_b32_u4kv cfg3_rank; do
for d in 1 0; do
  DIHIP_DEFER_RMSNORM=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5r/prof_${w}_$d -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 8 --warmup 2 --blocks 1 --no-cpu-baseline --no-graph --runner python --no-extra > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r5r/prof_${w}_$d -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats, $w, defer=$d"
  [ -n "$f" ] && grep dihip "$f" | head -12 | cut -d, -f1-4 | cut -c1-200 && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r5r/kernel_stats_${w}_defer$d.csv
  find $GRAFT_REPO_ROOT/gpurun_out/r5r/prof_${w}_$d -name "*.csv" -size +2M -delete
done
done
} 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r5r/log.txt
