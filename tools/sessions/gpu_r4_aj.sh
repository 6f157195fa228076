#!/bin/bash
# round 4, call aj: PMC traffic passes of the four workloads on the final kernel sources (comment-only edits since r04zzzz change the hash)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r04zzzzz
for w in int4_b1 int4_b32_u4kv cfg3_rank int8_b1; do
  bash tools/gpu_pmc.sh r04zzzzz/pmc $w > gpurun_out/r04zzzzz/pmc_$w.log 2>&1
  grep -c "FETCH_SIZE\|WRITE_SIZE" gpurun_out/r04zzzzz/pmc_$w.log
done
ls gpurun_out/r04zzzzz/pmc/*.csv
