#!/bin/bash
# r6 g: error word read at sync + recovery (fault injection), occupancy check; attention block + host runner suites
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_attn_block.py tests/test_gpu_attn_merge_stress.py tests/test_gpu_host_runner.py tests/test_gpu_rccl_one_rank.py -q -x --timeout 900 2>&1 | tail -15 | tee $OUT/pytest.log
