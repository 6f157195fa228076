#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_host_runner.py -x -q -k "f16" -s > gpurun_out/i_f16.log 2>&1; echo "rc=$?" >> gpurun_out/i_f16.log
tail -30 gpurun_out/i_f16.log
