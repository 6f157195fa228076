#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6t
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x --deselect tests/test_gpu_parity_depth.py 2>&1 | tail -30 | tee $OUT/pytest.log
