#!/bin/bash
# round 5, call b: timeline of the fused attention block (trace build) + parity re-run after the grid fix
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5b
{
timeout 600 python -m pytest tests/test_gpu_attn_block.py -q -m gpu -x --timeout 300 2>&1 | tail -5
DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/attn_block_trace.py 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r5b/log.txt
