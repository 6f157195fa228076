#!/bin/bash
# round 5, call v: regression run after the deferred RMSNorm: GEMM / decoder / host-runner / parity-depth suites
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5v
timeout 2400 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_decoder.py tests/test_gpu_host_runner.py tests/test_gpu_parity_depth.py tests/test_gpu_f16_small_batch.py -q -m gpu --timeout 900 -s 2>&1 | grep -E "FULL DEPTH|\[int4_b1\]|\[int4_b32_u4kv\]|\[cfg3_rank\]|passed|failed|Error|assert|FAILED" | cut -c1-600 | tee gpurun_out/r5v/log.txt | tail -40
