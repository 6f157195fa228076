#!/bin/bash
# r6 b: attention-block timeline with finer stamps inside the tile (trace build) + 1-rank RCCL tests
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_rccl_one_rank.py -q -x --timeout 300 2>&1 | tail -8 > $OUT/pytest.log
cat $OUT/pytest.log
DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/attn_block_trace.py 2>&1 | tee $OUT/trace_7b.txt
