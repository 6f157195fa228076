#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_host_ops.py -q --timeout 600 2>&1 | tail -30 | tee $OUT/pytest.log
