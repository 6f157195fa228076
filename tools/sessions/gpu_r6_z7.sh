#!/bin/bash
# r6 z7: int8 block: how much of the 16-chunk qkv share goes out before the RMSNorm prologue (4 / 8 / 12 / 16 slots)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6z7
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for V in product w8e4 w8e12 w8e16; do
  D=""; [ $V != product ] && D=$PWD/dash-infer_amd/lib/$V
  DIHIP_LIB_DIR=$D timeout 300 python bench.py --workload int8_b1 --no-extra --no-cpu-baseline --runner python --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('int8_b1 $V', d['value'], d['ms_per_step'], d.get('kernels_us',{}).get('attn_block_qkv_attention_o'))" | tee -a $OUT/sweep.txt
done
done
