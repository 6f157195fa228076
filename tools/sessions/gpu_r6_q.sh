#!/bin/bash
# r6 q: exact poll width + two-phase wide merge + polled-only block (177 VGPRs): tests (incl. long histories -> > 20 splits), timing, timeline
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_attn_block.py tests/test_gpu_host_runner.py -q -x --timeout 600 2>&1 | tail -4 | tee $OUT/pytest.log
for rep in 1 2 3; do
  r=$(timeout 300 python tools/attn_block_trace.py 2>&1 | grep "one launch" | sed 's/.*: *//; s/ us per.*//')
  echo "rep $rep product -> $r" | tee -a $OUT/sweep.txt
done
HIST=3900 timeout 300 python tools/attn_block_trace.py 2>&1 | grep "us per layer\|served" | tee -a $OUT/sweep.txt
DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/attn_block_trace.py 2>&1 | tee $OUT/trace_7b.txt | tail -30
timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('kernels_us'))" | tee -a $OUT/sweep.txt
