#!/bin/bash
# r6 j: TP in the C++ layer: device-resident tail, graph replay at TP 2/4/8 batch 1 / 16
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_host_runner.py tests/test_gpu_sampling.py -q --timeout 900 2>&1 | tail -25 | tee $OUT/pytest.log
