#!/bin/bash
# r6 p: attention block: poll width 20 vs 24, sleeps in the record / q poll loops -- graph-timed 8-layer tool, two rounds
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_attn_block.py -q -x --timeout 600 2>&1 | tail -3 | tee $OUT/pytest.log
for rep in 1 2 3; do
for V in "" ab_mb24 ab_sm0 ab_sm6 ab_sq0 ab_sq4; do
  D=$PWD/dash-infer_amd/lib${V:+/$V}
  r=$(DIHIP_LIB_DIR=$D timeout 300 python tools/attn_block_trace.py 2>&1 | grep "one launch" | sed 's/.*: *//; s/ us per.*//')
  echo "rep $rep ${V:-product} -> $r" | tee -a $OUT/sweep.txt
done
done
