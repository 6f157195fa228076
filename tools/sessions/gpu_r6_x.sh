#!/bin/bash
# r6 x: GEMV workgroups take the first block ids; the first V tile goes to LDS before the wait for q -- A/B against variants, block tests
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6x
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for V in product attnfirst vlate; do
    D=""; [ $V != product ] && D=$PWD/dash-infer_amd/lib/$V
    r=$(DIHIP_LIB_DIR=$D timeout 300 python tools/attn_block_trace.py 2>&1 | grep "one launch" | sed 's/.*: *//; s/ us per.*//')
    echo "rep $rep $V -> $r" | tee -a $OUT/sweep.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_attn_block.py -q --timeout 600 2>&1 | tail -5 | tee $OUT/pytest.log
for V in product attnfirst vlate; do
D=""; [ $V != product ] && D=$PWD/dash-infer_amd/lib/$V
DIHIP_LIB_DIR=$D timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $V', d['value'], d['ms_per_step'])" | tee -a $OUT/sweep.txt
done
make -C dash-infer_amd/csrc trace -j16 2>&1 | grep -E "error" | head
DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/attn_block_trace.py 2>&1 | tee $OUT/trace_7b.txt | tail -34
