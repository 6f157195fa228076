#!/bin/bash
# round 4, call q: in-situ A/B of the K-slice kernel's reducer wave (DIHIP_KSLICE_SPARE_WAVE) on the two small-batch workloads
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
{
for rep in 1 2; do for sw in 0 1; do for w in int4_b32_u4kv cfg3_rank; do
  DIHIP_KSLICE_SPARE_WAVE=$sw timeout 300 python bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('spare=$sw', '$w', 'tok/s', d['value'], 'ms', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if 'gate' in k or 'down' in k})
"
done; done; done
} 2>&1 | tee gpurun_out/r4q_spare_wave.txt
