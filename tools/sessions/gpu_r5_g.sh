#!/bin/bash
# round 5, call g: the reference converter's serialized graph through the C++ layer; host-layer suites after the renames
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5g
{
timeout 1200 python -m pytest tests/test_gpu_host_runner.py tests/test_gpu_host_graph.py tests/test_gpu_host_ops.py tests/test_gpu_sampling.py -q -m gpu --timeout 600 2>&1 | tail -8
timeout 400 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-extra 2>gpurun_out/r5g/bench.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('headline tok/s', d['value'], 'ms', d['ms_per_step'], d['runner'][:40])"
} 2>&1 | tee gpurun_out/r5g/log.txt
