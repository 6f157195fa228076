#!/bin/bash
# r6 v: the polled merge with four lanes per item (ceil(splits / 4) records per lane and pass) against one lane per item; both attention forms
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6v
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for W in 0 1; do
    for V in quad noquad; do
      D=""; [ $V = noquad ] && D=$PWD/dash-infer_amd/lib/noquad
      r=$(DIHIP_LIB_DIR=$D DIHIP_ATTN_WIDE=$W timeout 300 python tools/attn_block_trace.py 2>&1 | grep "one launch" | sed 's/.*: *//; s/ us per.*//')
      echo "rep $rep WIDE=$W $V -> $r" | tee -a $OUT/sweep.txt
    done
  done
done
for W in 0 1; do
DIHIP_ATTN_WIDE=$W timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench quad WIDE=$W', d['value'], d['ms_per_step'])" | tee -a $OUT/sweep.txt
done
timeout 900 python -m pytest tests/test_gpu_attn_block.py -q --timeout 600 2>&1 | tail -12 | tee $OUT/pytest.log
