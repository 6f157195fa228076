#!/bin/bash
# r6 z6: rocprofv3 kernel statistics of int8_b1 (BASELINE configs[1]) with the int8 form of the attention block, and with it off
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6z6
mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for P in 1 0; do
  DIHIP_ATTN_BLOCK_W8=$P timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -o p$P -- python $GRAFT_REPO_ROOT/bench.py --workload int8_b1 --no-cpu-baseline --no-extra > $OUT/prof_$P.json 2> $OUT/prof_$P.err
  f=$(find $OUT/prof_$P -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/bench_int8_b1_kernel_stats_w8block$P.csv
  echo "== W8 block $P"; grep dihip $f | head -6 | cut -d, -f1-4 | cut -c1-150
  tail -1 $OUT/prof_$P.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  find $OUT/prof_$P -name "*.csv" -size +2M -delete
done
