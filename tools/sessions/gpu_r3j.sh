#!/bin/bash
# round 3, call J: rocprofv3 kernel statistics of the context phase (profiles/r03j_bench_prefill_2048_kernel_stats.csv)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3j; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o pf -- python $ROOT/bench.py --workload prefill_2048 --steps 4 --warmup 1 > $OUT/prof_prefill.json 2> $OUT/prof_prefill.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_prefill_2048_kernel_stats.csv && grep dihip $OUT/bench_prefill_2048_kernel_stats.csv | head -16 | cut -c1-220
find $OUT/prof -name "*.csv" -size +2M -delete
