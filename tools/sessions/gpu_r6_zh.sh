#!/bin/bash
# r6 zh: the MLP half as one launch (dihip_decode_mlp_block, opt-in) at the TP = 8 rank shape of the 7B model, where both GEMVs are a few MB and the
# launches' fixed costs dominate -- Python runner, A/B in one call
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6zh
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for W in tp8_rank_7b int4_b1; do
for V in 0 1; do
  DIHIP_DECODER_MLP_BLOCK=$V timeout 300 python bench.py --workload $W --no-extra --no-cpu-baseline --runner python --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W mlp_block=$V', d['value'], d['ms_per_step'])" | tee -a $OUT/sweep.txt
done
done
done
