#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity_depth.py > gpurun_out/f_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/f_pytest_gpu.log
tail -15 gpurun_out/f_pytest_gpu.log
