#!/bin/bash
# round 4, call aa: uint4 attention with P' as one bf16 (no lo pass): accuracy against the oracle + timing
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kv_attn.py tests/test_gpu_decoder.py tests/test_gpu_host_runner.py -k "not qwen7b and not depth" -q -x -m gpu 2>&1 | tail -6
{
python tools/attn_step_bench.py 2>&1 | grep "us/layer"
timeout 300 python bench.py --workload int4_b32_u4kv --steps 16 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('int4_b32_u4kv', 'tok/s', d['value'], 'ms', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})
"
} 2>&1 | tee gpurun_out/r4aa_u4_p_single.txt
