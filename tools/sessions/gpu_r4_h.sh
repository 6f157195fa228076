#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_attn_merge_stress.py tests/test_gpu_host_ops.py -x -q > gpurun_out/h_stress.log 2>&1; echo "rc=$?" >> gpurun_out/h_stress.log
tail -25 gpurun_out/h_stress.log
