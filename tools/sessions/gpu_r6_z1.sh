#!/bin/bash
# r6 z1: does a decode GEMV start faster on weights the launch before it pulled into the Infinity Cache? (tools/mall_prefetch_bench.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6z1
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for M in cold full head; do
  for H in 16; do
    MODE=$M HEAD_MB=$H timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$M -o $M -- python $GRAFT_REPO_ROOT/tools/mall_prefetch_bench.py > $OUT/run_$M.log 2>&1
    f=$(find $OUT/prof_$M -name "*kernel_stats.csv" | head -1)
    echo "== $M" | tee -a $OUT/summary.txt
    grep "dihip" $f | head -6 | cut -d, -f1-4 | cut -c1-150 | tee -a $OUT/summary.txt
    find $OUT/prof_$M -name "*.csv" -size +2M -delete
  done
done
