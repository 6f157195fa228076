#!/bin/bash
# round 4, call u: the reworked one-launch uint4 step (wave 0 quantises K and V, no barrier): tests + isolated timing + workload A/B
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kv_attn.py -k "u4_decode_step or fused_rope" -q -x -m gpu 2>&1 | tail -4
{
python tools/attn_step_bench.py 2>&1 | grep "us/layer"
for f in 1 0; do
  DIHIP_ATTN_U4_FUSED=$f timeout 300 python bench.py --workload int4_b32_u4kv --steps 16 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('u4_fused=$f', 'tok/s', d['value'], 'ms', d['ms_per_step'])
"
done
} 2>&1 | tee gpurun_out/r4u_u4_step.txt
