#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6o
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_tp_loopback.py -q --timeout 900 -k "beside_the_next or serialized_export" 2>&1 | tail -15 | tee $OUT/pytest.log
