#!/bin/bash
# round 4, call t: the uint4 decode step in one launch (Rotary + quantising append + attention): tests, then the
# batch-32 workload with the one-launch form and with the append launch (DIHIP_ATTN_U4_FUSED=0)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kv_attn.py tests/test_gpu_host_runner.py tests/test_gpu_decoder.py -k "(u4_decode_step or fused_rope or frag32 or bit_identical or greedy or batch32) and not depth and not qwen7b" -q -x -m gpu 2>&1 | tail -6
{
for rep in 1 2; do for f in 1 0; do
  DIHIP_ATTN_U4_FUSED=$f timeout 300 python bench.py --workload int4_b32_u4kv --steps 16 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('u4_fused=$f', 'tok/s', d['value'], 'ms', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})
"
done; done
} 2>&1 | tee gpurun_out/r4t_u4_step.txt
