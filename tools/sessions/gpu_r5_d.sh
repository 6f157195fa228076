#!/bin/bash
# round 5, call d: split count of the attention inside the block (fewer attention workgroups = more GEMV workgroups, fewer records to merge)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5d
{
for ns in 0 9 11 13 17; do
  echo "== DIHIP_ATTN_NSPLITS=$ns"
  DIHIP_ATTN_NSPLITS=$ns timeout 300 python tools/attn_block_trace.py 2>&1 | grep -E "launch|served"
done
} 2>&1 | tee gpurun_out/r5d/log.txt
