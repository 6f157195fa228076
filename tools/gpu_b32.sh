#!/bin/bash
# batch-32 loop: kernel microbenches + b32 bench
export LD_LIBRARY_PATH=$PWD/dash-infer_amd/lib:/opt/rocm/lib
./tools/gemv_bench 4 128 32 | grep -v "warm\|pre-trace\|trace launch\|plan:\|lm_head"
for w in 1 2 4; do echo "WGS_PER_CU=$w"; DIHIP_ATTN_WGS_PER_CU=$w timeout 300 python bench.py --workload int4_b32_u4kv --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tok/s', d['value'], 'ms', d['ms_per_step'])
for k,v in d['kernels'].items(): print('  %-28s %8.2f us' % (k, v['avg_us']))
"; done
