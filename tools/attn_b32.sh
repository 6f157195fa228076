#!/bin/bash
# batch-32 u4 attention: split-count sweep
export LD_LIBRARY_PATH=$PWD/dash-infer_amd/lib:/opt/rocm/lib
for w in 1 2 4 8; do echo "NSPLITS=$w"; DIHIP_ATTN_NSPLITS=$w timeout 300 python bench.py --workload int4_b32_u4kv --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tok/s', d['value'], 'ms', d['ms_per_step'], 'attn', d['kernels']['rope_append_span_attention']['avg_us'])
"; done
