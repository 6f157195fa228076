#!/bin/bash
# r6 zk: the generic GEMM with 2 column tiles per wave instead of 4 for the unquantised lm_head of batched decode (594 -> 1188 column blocks over 512
# resident slots: a finer tail), A/B through a diagnostics switch
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6zk
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for W in int4_b32_u4kv cfg3_rank cfg5_moe; do
for T in 0 2; do
  DIHIP_DENSE_NT=$T timeout 300 python bench.py --workload $W --no-extra --no-cpu-baseline --runner python --steps 16 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W dense_nt=$T', d['value'], d['ms_per_step'], (d.get('kernels_us') or {}).get('lm_head'), d.get('last_ids'))" | tee -a $OUT/sweep.txt
done
done
done
