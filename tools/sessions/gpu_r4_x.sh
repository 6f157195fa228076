#!/bin/bash
# round 4, call x: K-slice GEMM with the reducer's LDS reads batched: stand-alone timing + the two workloads + parity
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -k "kslice" -q -x -m gpu 2>&1 | tail -3
{
for m in 32 16; do
  printf "M=%d W4  " $m
  SHAPE=gate_up_swiglu LD_LIBRARY_PATH=dash-infer_amd/lib timeout 120 ./tools/gemv_bench 4 128 $m 5 2>&1 | grep -v "warm-up" | tail -1 | sed 's/.*avg/avg/'
done
for rep in 1 2; do for w in cfg3_rank int4_b32_u4kv; do
  timeout 300 python bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$w', 'tok/s', d['value'], 'ms', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if 'gate' in k or 'down' in k})
"
done; done
} 2>&1 | tee gpurun_out/r4x_kslice_reduce.txt
