#!/bin/bash
# round 3, call AB: split count of the decode attention at configs[3]'s rank shape (batch 16, 8 query / 1 KV head, 4096 tokens)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3ab; mkdir -p $OUT; cd $ROOT
bench() {
  local name=$1; local w=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d['kernels']['rope_append_span_attention']['avg_us'])"
}
bench cfg3_s32 cfg3_rank X=1
bench cfg3_s16 cfg3_rank DIHIP_ATTN_NSPLITS=16
bench cfg3_s8 cfg3_rank DIHIP_ATTN_NSPLITS=8
bench cfg3_s4 cfg3_rank DIHIP_ATTN_NSPLITS=4
bench moe_s cfg5_moe DIHIP_ATTN_NSPLITS=4
