#!/bin/bash
# round 4, call v: what-if builds of the one-launch uint4 step (u4x1: no Rotary of q, u4x2: no new token, u4x3: neither)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for v in "" u4x1 u4x2 u4x3; do
  echo "== ${v:-product}"
  DIHIP_LIB_DIR=${v:+$GRAFT_REPO_ROOT/dash-infer_amd/lib/$v} python tools/attn_step_bench.py 2>&1 | grep "B=32" | grep -v "append"
  DIHIP_LIB_DIR=${v:+$GRAFT_REPO_ROOT/dash-infer_amd/lib/$v} python tools/attn_step_bench.py 2>&1 | grep "B=32" | grep "one-launch"
done 2>&1 | tee gpurun_out/r4v_u4_step_whatif.txt
