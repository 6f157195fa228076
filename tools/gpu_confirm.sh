#!/bin/bash
# Short confirmation session after a host-side change: the GPU suite, the headline bench line and the stamped PMC passes.
#   gpurun --timeout 1500 -- 'TAG=r03zzz bash tools/gpu_confirm.sh'
TAG=${TAG:-confirm}
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -3 | cut -c1-300 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
timeout 400 python bench.py > $OUT/bench_int4_b1.json 2> $OUT/bench_int4_b1.err
python -c "
import json; d=json.load(open('$OUT/bench_int4_b1.json')); print('int4_b1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
bash tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1; tail -4 $OUT/pmc.log | cut -c1-160
