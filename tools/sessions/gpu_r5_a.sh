#!/bin/bash
# round 5, call a: the fused attention block (dihip_decode_attn_block) -- parity against the three launches it replaces, then
# the headline step with the block off / on (Python runner, graph replay)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5a
{
timeout 600 python -m pytest tests/test_gpu_attn_block.py -q -m gpu -x --timeout 300 2>&1 | tail -15
for blk in 0 1; do
  DIHIP_DECODER_ATTN_BLOCK=$blk timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extra --runner python 2>gpurun_out/r5a/bench_blk$blk.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('attn_block=$blk', 'tok/s', d['value'], 'ms', d['ms_per_step'], d.get('blocks'), {k: v['avg_us'] for k, v in d['kernels'].items()})
"
  tail -3 gpurun_out/r5a/bench_blk$blk.err
done
} 2>&1 | tee gpurun_out/r5a/log.txt
