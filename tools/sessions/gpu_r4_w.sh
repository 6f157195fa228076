#!/bin/bash
# round 4, call w: panel kernel with temporal activation loads: parity + the two small-batch workloads
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py -k "panel or frag or kslice or residual or tp8" -q -x -m gpu 2>&1 | tail -4
{
for rep in 1 2; do for w in cfg3_rank int4_b32_u4kv; do
  timeout 300 python bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$w', 'tok/s', d['value'], 'ms', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})
"
done; done
} 2>&1 | tee gpurun_out/r4w_panel_temporal_x.txt
