#!/bin/bash
# r6 y: timeline by split index (which attention workgroup is late?), block tests, timing
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6y
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_attn_block.py tests/test_gpu_kv_attn.py -q --timeout 600 2>&1 | tail -5 | tee $OUT/pytest.log
for rep in 1 2 3; do
  r=$(timeout 300 python tools/attn_block_trace.py 2>&1 | grep "one launch" | sed 's/.*: *//; s/ us per.*//')
  echo "rep $rep -> $r" | tee -a $OUT/sweep.txt
done
make -C dash-infer_amd/csrc trace -j16 2>&1 | grep -E "error" | head
BY_SPLIT=1 DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/attn_block_trace.py 2>&1 | tee $OUT/trace_7b.txt | tail -40
timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])" | tee -a $OUT/sweep.txt
