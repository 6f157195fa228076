#!/bin/bash
# round 4, call ab: what-if builds of the uint4 attention kernel: u4x4 loads only, u4x8 no V^T.P' half
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for v in "" u4x4 u4x8; do
  echo "== ${v:-product}"
  DIHIP_LIB_DIR=${v:+$GRAFT_REPO_ROOT/dash-infer_amd/lib/$v} python tools/attn_step_bench.py 2>&1 | grep "attention alone"
done 2>&1 | tee gpurun_out/r4ab_u4_attn_whatif.txt
