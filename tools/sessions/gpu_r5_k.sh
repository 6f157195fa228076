#!/bin/bash
# round 5, call k: timeline of the MLP block (trace build) + parity
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5k
{
timeout 900 python -m pytest tests/test_gpu_mlp_block.py -q -m gpu -x --timeout 300 2>&1 | tail -4
DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/mlp_block_trace.py 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r5k/log.txt
