#!/bin/bash
# r6 zj: the generic GEMM's split-K target (make_plan: 2.5 workgroups per CU = 640) against multiples of the resident slots, on the workloads whose
# lm_head runs on it (cfg3_rank: 75 column blocks x 8 splits = 600 workgroups; int4_b32_u4kv: 594 column blocks, no split)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6zj
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for W in cfg3_rank int4_b32_u4kv; do
for T in 0 512 1024 1536; do
  DIHIP_GEMM_TARGET_BLOCKS=$T timeout 300 python bench.py --workload $W --no-extra --no-cpu-baseline --runner python --steps 16 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W target=$T', d['value'], d['ms_per_step'], (d.get('kernels_us') or {}).get('lm_head'))" | tee -a $OUT/sweep.txt
done
done
done
