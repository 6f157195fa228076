#!/bin/bash
# timing-only bound: the prefill GEMM skeleton without expansion / fix-up / row sums (results are wrong on purpose)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3r; mkdir -p $OUT; cd $ROOT
for v in pf_fake; do
  DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/$v timeout 300 python bench.py --workload prefill_2048 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python -c "
import json; d=json.load(open('$OUT/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['gemms'])"
done
