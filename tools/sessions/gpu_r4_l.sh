#!/bin/bash
# round 4, call l: the MoE layer through the operator API (fusion pass -> DihipMoeBlock), + the MoE host operators after the header split
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_host_runner.py tests/test_gpu_host_ops.py tests/test_gpu_decoder.py -k "not qwen7b and not depth" -q -x -m gpu 2>&1 | tail -15 > gpurun_out/r4l_pytest.log
tail -15 gpurun_out/r4l_pytest.log
