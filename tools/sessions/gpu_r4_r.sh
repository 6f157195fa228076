#!/bin/bash
# round 4, call r: with temporal activation loads, does the K-slice kernel now also win on the small matrices (qkv, o)?
# DIHIP_GEMM_KSLICE=2 forces it for every eligible shape; 1 = default plan (>= 24 MB only)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
{
for ks in 1 2; do for w in int4_b32_u4kv cfg3_rank; do
  DIHIP_GEMM_KSLICE=$ks timeout 300 python bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('kslice=$ks', '$w', 'tok/s', d['value'], 'ms', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})
"
done; done
} 2>&1 | tee gpurun_out/r4r_kslice_all.txt
