#!/bin/bash
# round 5, call ag: the operator-layer suites after the last host-side changes (P2P branch of the AllReduce operator, weight file reader)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5ag
timeout 140 python -m pytest tests/test_gpu_host_ops.py tests/test_gpu_host_graph.py -q -m gpu --timeout 100 -x 2>&1 | tail -4 | cut -c1-300 | tee gpurun_out/r5ag/log.txt
