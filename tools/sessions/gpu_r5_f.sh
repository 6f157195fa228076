#!/bin/bash
# round 5, call f: quantised-KV codec vs the reference's device code (after the comparison of zero-points by value)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5f
timeout 600 python -m pytest tests/test_gpu_kv_codec_ref.py -q -m gpu --timeout 300 -s 2>&1 | grep -E "kv codec|passed|failed|Error|Mismatch" | tee gpurun_out/r5f/log.txt
