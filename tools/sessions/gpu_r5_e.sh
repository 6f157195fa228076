#!/bin/bash
# round 5, call e: the quantised-KV codec against the reference's own device code; new bench workload; rocprof-in-run
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5e
{
timeout 600 python -m pytest tests/test_gpu_kv_codec_ref.py -q -m gpu --timeout 300 -s 2>&1 | tail -25
timeout 300 python bench.py --workload tp8_rank_7b --steps 32 --warmup 4 --no-cpu-baseline --no-extra 2>gpurun_out/r5e/tp8.err > gpurun_out/r5e/tp8.json
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r5e/tp8.json") if l.startswith('{')][-1])
print('tp8_rank_7b tok/s', d['value'], 'ms', d['ms_per_step'], d['step_hbm'], {k: v['avg_us'] for k, v in d['kernels'].items()}, d['python_runner'])
PY
tail -3 gpurun_out/r5e/tp8.err
DIHIP_DECODER_ATTN_BLOCK=0 timeout 300 python bench.py --workload tp8_rank_7b --steps 32 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('tp8_rank_7b chain: tok/s', d['value'], 'ms', d['ms_per_step'])"
} 2>&1 | tee gpurun_out/r5e/log.txt
