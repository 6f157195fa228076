#!/bin/bash
# round 5, call ac: the tensor-parallel loop-back suite (Python runner + the C++ operator layer) after the weight-slicing refactor
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5ac
timeout 900 python -m pytest tests/test_gpu_tp_loopback.py -q -m gpu --timeout 600 2>&1 | tail -5 | tee gpurun_out/r5ac/log.txt
