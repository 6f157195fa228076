#!/bin/bash
# r6 z5: the attention block for int8 per-channel weights (BASELINE configs[1]): A/B with DIHIP_ATTN_BLOCK_W8=0, both runners; int4 unchanged?
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6z5
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for P in 1 0; do
  for R in host python; do
  DIHIP_ATTN_BLOCK_W8=$P timeout 300 python bench.py --workload int8_b1 --no-extra --no-cpu-baseline --runner $R --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('int8_b1 W8BLOCK=$P $R', d['value'], d['ms_per_step'], d.get('kernels_us'))" | tee -a $OUT/sweep.txt
  done
done
done
timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('int4_b1', d['value'], d['ms_per_step'], d.get('kernels_us'))" | tee -a $OUT/sweep.txt
