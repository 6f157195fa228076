#!/bin/bash
# r6 z2: the attention block pulls the head of the gate / up weights on-die while it waits (dihip_decode_attn_block_pf): A/B with DIHIP_ATTN_BLOCK_PREFETCH=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6z2
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_attn_block.py tests/test_gpu_host_runner.py -q -x --timeout 600 2>&1 | tail -4 | tee $OUT/pytest.log
for rep in 1 2; do
for P in 1 0; do
  for R in host python; do
  DIHIP_ATTN_BLOCK_PREFETCH=$P timeout 300 python bench.py --no-extra --no-cpu-baseline --runner $R --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench PREFETCH=$P $R', d['value'], d['ms_per_step'], d.get('kernels_us'))" | tee -a $OUT/sweep.txt
  done
done
done
export TMPDIR=/tmp; cd /tmp
for P in 1 0; do
  DIHIP_ATTN_BLOCK_PREFETCH=$P timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -o p$P -- python $GRAFT_REPO_ROOT/bench.py --workload int4_b1 --no-cpu-baseline --no-extra > $OUT/prof_$P.json 2> $OUT/prof_$P.err
  f=$(find $OUT/prof_$P -name "*kernel_stats.csv" | head -1)
  echo "== PREFETCH=$P" | tee -a $OUT/sweep.txt
  grep dihip $f | head -4 | cut -d, -f1-4 | cut -c1-150 | tee -a $OUT/sweep.txt
  find $OUT/prof_$P -name "*.csv" -size +2M -delete
done
