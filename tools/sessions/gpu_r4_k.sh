#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_host_runner.py tests/test_gpu_host_graph.py -x -q > gpurun_out/k_runner.log 2>&1; echo "rc=$?" >> gpurun_out/k_runner.log
tail -20 gpurun_out/k_runner.log
