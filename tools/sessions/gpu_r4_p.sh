#!/bin/bash
# round 4, call p: K-slice GEMM after (a) temporal activation loads, (b) the slice-less wave as reducer: parity, stand-alone
# timing with the reducer wave on / off, then the two small-batch workloads
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -k "kslice or fused_norm or residual" -q -x -m gpu 2>&1 | tail -4
{
for sw in 0 1; do
  for m in 32 16; do
    printf "spare_wave=%d M=%d W4  " $sw $m
    DIHIP_KSLICE_SPARE_WAVE=$sw SHAPE=gate_up_swiglu LD_LIBRARY_PATH=dash-infer_amd/lib timeout 120 ./tools/gemv_bench 4 128 $m 5 2>&1 | grep -v "warm-up" | tail -1 | sed 's/.*avg/avg/'
  done
  printf "spare_wave=%d M=32 W8  " $sw
  DIHIP_KSLICE_SPARE_WAVE=$sw SHAPE=gate_up_swiglu LD_LIBRARY_PATH=dash-infer_amd/lib timeout 120 ./tools/gemv_bench 8 -1 32 5 2>&1 | grep -v "warm-up" | tail -1 | sed 's/.*avg/avg/'
done
for w in int4_b32_u4kv cfg3_rank; do
  timeout 300 python bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$w', 'tok/s', d['value'], 'ms', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})
"
done
} 2>&1 | tee gpurun_out/r4p_kslice.txt
