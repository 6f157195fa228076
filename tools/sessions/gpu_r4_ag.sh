#!/bin/bash
# round 4, call ag: the fusion pass after it learnt the converter's own arities: host runner / host graph tests
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_host_runner.py tests/test_gpu_host_graph.py -q -x -m gpu 2>&1 | tail -5 | tee gpurun_out/r4ag_host_tests.txt
