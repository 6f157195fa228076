#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_attn_block.py -q --timeout 600 2>&1 | tail -12 | tee $OUT/pytest.log
