#!/bin/bash
# K-slice vs panel kernel at batch 32: parity tests, then the b32 bench with either kernel
mkdir -p gpurun_out/ksl
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu -x --timeout 600 > gpurun_out/ksl/tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/ksl/tests.log
for k in 0 1; do echo "KSLICE=$k"; DIHIP_GEMM_KSLICE=$k timeout 300 python bench.py --workload int4_b32_u4kv --steps 16 --warmup 4 --no-cpu-baseline 2>gpurun_out/ksl/bench$k.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tok/s', d['value'], 'ms', d['ms_per_step'])
for k,v in d['kernels'].items(): print('  %-28s %8.2f us' % (k, v['avg_us']))
"; done
