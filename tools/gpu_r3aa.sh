#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3aa; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_glue.py -m gpu -q -x 2>&1 | tail -2 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x -k "real_width or batched or greedy" 2>&1 | tail -2 | cut -c1-300
for w in cfg3_rank; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], {k: v['avg_us'] for k, v in d.get('kernels', {}).items()})"
done
