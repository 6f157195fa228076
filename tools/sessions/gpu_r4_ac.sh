#!/bin/bash
# round 4, call ac: split count of the uint4 decode attention at batch 32 (workgroups per CU the plan aims at)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for n in 1 2 3 4 6; do
  echo "== DIHIP_ATTN_WGS_PER_CU=$n"
  DIHIP_ATTN_WGS_PER_CU=$n python tools/attn_step_bench.py 2>&1 | grep "B=32" | grep -v append
done 2>&1 | tee gpurun_out/r4ac_u4_attn_splits.txt
