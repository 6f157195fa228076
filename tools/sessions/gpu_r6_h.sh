#!/bin/bash
# r6 h: TP ranks split one serialized export at load; the rest of the loop-back suite
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_tp_loopback.py -q -x --timeout 900 -k "serialized_export" 2>&1 | tail -25 | tee $OUT/pytest.log
