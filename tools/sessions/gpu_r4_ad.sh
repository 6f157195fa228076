#!/bin/bash
# round 4, call ad: after the last edit of a kernel source (what-if hooks in span_attn.hip, compiled out): attention / decoder / host
# runner tests, the batch-32 line with the one-launch step in its per-kernel table, PMC passes for the four workloads
TAG=r04ad
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests/test_gpu_kv_attn.py tests/test_gpu_decoder.py tests/test_gpu_host_runner.py tests/test_gpu_host_ops.py -k "not qwen7b and not depth" -q -m gpu 2>&1 | tail -3 | tee gpurun_out/$TAG/pytest_subset.log
for w in int4_b32_u4kv cfg3_rank; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/$TAG/bench_$w.json 2> gpurun_out/$TAG/bench_$w.err
  python -c "
import json
d = json.load(open('gpurun_out/$TAG/bench_$w.json'))
print('$w', d['value'], d['ms_per_step'], d['step_hbm']['frac_of_peak'], {k: v['avg_us'] for k, v in d['kernels'].items()}, d['roofline'].get('traffic'), d['roofline'].get('traffic_source'))
"
done
for w in int4_b1 int4_b32_u4kv cfg3_rank int8_b1; do
  bash tools/gpu_pmc.sh $TAG/pmc $w > gpurun_out/$TAG/pmc_$w.log 2>&1
  tail -4 gpurun_out/$TAG/pmc_$w.log | cut -c1-140
done
