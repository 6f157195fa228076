#!/bin/bash
# round 5, call t: is the K-slice kernel's plain path slower with the deferred-norm consumer compiled in? (lib/ksl_nors: -DDIHIP_KSL_RS=0)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5t
{ for rep in 1 2; do
echo "== product build"; timeout 300 python tools/defer_norm_bench.py 32 2>&1 | grep -E "SwiGLU pair alone"
echo "== consumer compiled out"; DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/ksl_nors PLAIN_ONLY=1 timeout 300 python tools/defer_norm_bench.py 32 2>&1 | grep -E "SwiGLU pair alone|Error|error"
echo "== exchange arrays aliased (no LDS growth; results wrong)"; DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/ksl_tiny timeout 300 python tools/defer_norm_bench.py 32 2>&1 | grep -E "SwiGLU pair alone|Error|error"
done; } 2>&1 | tee gpurun_out/r5t/log.txt
