#!/bin/bash
# round 4, call al: rocprofv3 --kernel-trace --stats of the two small-batch workloads (eager launches, python runner) on the final tree
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r4al
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for w in int4_b32_u4kv cfg3_rank; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4al/$w -o st -- python $R/bench.py --workload $w --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-graph --runner python --no-extra > $R/gpurun_out/r4al/$w.json 2> $R/gpurun_out/r4al/$w.err
  f=$(find $R/gpurun_out/r4al/$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $R/gpurun_out/r4al/${w}_kernel_stats.csv && head -14 "$f" | cut -c1-150
done
find $R/gpurun_out/r4al -name "*.csv" -size +4M -delete
