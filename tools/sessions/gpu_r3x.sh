#!/bin/bash
# round 3, call X: shared-expert down projection from FRAG32 activations (MoE step)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3x; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x -k "moe" 2>&1 | tail -2 | cut -c1-300
for m in 1 0; do
  DIHIP_MOE_ACT_FRAG=$m timeout 300 python bench.py --workload cfg5_moe --no-cpu-baseline > $OUT/bench_frag$m.json 2> $OUT/bench_frag$m.err
  python -c "
import json; d=json.load(open('$OUT/bench_frag$m.json')); print('cfg5_moe act_frag=$m', d['value'], d['ms_per_step'])"
done
