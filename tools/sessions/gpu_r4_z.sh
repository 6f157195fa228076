#!/bin/bash
# round 4, call z: split-K panel plan for the skinny matrices (qkv, o) at small batch: DIHIP_PANEL_MIN_KTPS relaxes the plan's
# "at least 8 k-tiles per slice" rule
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
{
for k in 8 4 2; do for w in int4_b32_u4kv cfg3_rank; do
  DIHIP_PANEL_MIN_KTPS=$k timeout 300 python bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('min_ktps=$k', '$w', 'tok/s', d['value'], 'ms', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})
"
done; done
} 2>&1 | tee gpurun_out/r4z_panel_min_ktps.txt
