#!/bin/bash
# round 5, call c: the attention block through the C++ operator layer + headline bench (C++ runner value, Python runner beside it)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5c
{
timeout 900 python -m pytest tests/test_gpu_attn_block.py tests/test_gpu_host_runner.py tests/test_gpu_host_ops.py tests/test_gpu_decoder.py -q -m gpu --timeout 600 2>&1 | tail -8
for blk in 1; do
  DIHIP_DECODER_ATTN_BLOCK=$blk timeout 400 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-extra 2>gpurun_out/r5c/bench_blk$blk.err > gpurun_out/r5c/bench_blk$blk.json
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5c/bench_blk$blk.json") if l.startswith('{')][-1])
print('attn_block=$blk', 'tok/s', d['value'], 'ms', d['ms_per_step'], d.get('blocks'), d.get('extra', {}).keys())
print({k: v for k, v in d.items() if k in ('runner', 'runner_python', 'host_runner')})
PY
  tail -2 gpurun_out/r5c/bench_blk$blk.err
done
} 2>&1 | tee gpurun_out/r5c/log.txt
