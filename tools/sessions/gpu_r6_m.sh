#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_attn_block.py -q --timeout 600 2>&1 | tail -15 | tee $OUT/pytest.log
