#!/bin/bash
# round 5, call j: the MLP block (gate/up + SwiGLU + down in one launch): parity, then the headline with it off / on
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5j
{
timeout 900 python -m pytest tests/test_gpu_mlp_block.py -q -m gpu -x --timeout 300 2>&1 | tail -12
for mlp in 0 1; do
  DIHIP_DECODER_MLP_BLOCK=$mlp timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extra --runner python 2>gpurun_out/r5j/bench_mlp$mlp.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('mlp_block=$mlp', 'tok/s', d['value'], 'ms', d['ms_per_step'], d.get('blocks'))
"
  tail -2 gpurun_out/r5j/bench_mlp$mlp.err
done
} 2>&1 | tee gpurun_out/r5j/log.txt
