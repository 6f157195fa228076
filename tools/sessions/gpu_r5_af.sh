#!/bin/bash
# round 5, call af: the attention block on a rank that does not carry the residual (h_res = NULL)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5af
timeout 200 python -m pytest tests/test_gpu_attn_block.py -q -m gpu --timeout 150 -k "residual" 2>&1 | tail -12 | cut -c1-300 | tee gpurun_out/r5af/log.txt
