#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3i; mkdir -p $OUT; cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k "prefill_gemm or matches_oracle or reference_test" ) > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log | cut -c1-600
timeout 300 python bench.py --workload prefill_2048 --steps 8 --warmup 2 > $OUT/bench_prefill_2048.json 2> $OUT/bench_prefill.err; cat $OUT/bench_prefill_2048.json | cut -c1-1500; tail -3 $OUT/bench_prefill.err
DIHIP_GEMM_PREFILL=0 timeout 300 python bench.py --workload prefill_2048 --steps 4 --warmup 1 --layers 4 > $OUT/bench_prefill_old.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_prefill_old.json')); print('general kernel, 4 layers:', d['ms_per_step'], 'ms')"
timeout 300 python bench.py --workload prefill_2048 --steps 4 --warmup 1 --layers 4 > $OUT/bench_prefill_new4.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_prefill_new4.json')); print('prefill kernel, 4 layers:', d['ms_per_step'], 'ms')"
( time timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_prefill.py -m gpu -q -x -k "not full_depth" ) > $OUT/pytest_decoder.log 2>&1
tail -5 $OUT/pytest_decoder.log | cut -c1-400
