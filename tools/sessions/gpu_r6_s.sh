#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6s
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_attn_block.py -q -x --timeout 600 2>&1 | tail -3 | tee $OUT/pytest.log
for rep in 1 2 3; do
  r=$(timeout 300 python tools/attn_block_trace.py 2>&1 | grep "one launch" | sed 's/.*: *//; s/ us per.*//')
  echo "rep $rep product -> $r" | tee -a $OUT/sweep.txt
done
timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('kernels_us'))" | tee -a $OUT/sweep.txt
