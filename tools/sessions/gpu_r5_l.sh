#!/bin/bash
# round 5, call l: f16 on the small-batch / context-phase kernels
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5l
timeout 1500 python -m pytest tests/test_gpu_f16_small_batch.py "tests/test_gpu_decoder.py::test_f16_greedy_decode_matches_oracle" tests/test_gpu_gemm.py -q -m gpu --timeout 600 2>&1 | tail -15 | tee gpurun_out/r5l/log.txt
