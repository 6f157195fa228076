#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_host_runner.py -x -q > gpurun_out/d_sampling.log 2>&1; echo "rc=$?" >> gpurun_out/d_sampling.log
tail -30 gpurun_out/d_sampling.log
