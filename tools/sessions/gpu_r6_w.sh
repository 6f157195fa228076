#!/bin/bash
# r6 w: quad-lane polled merge + merge_order4 in every split merge + Rotary in the granule sweep: attention / runner / decoder suites, timing, timeline
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6w
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_attn_block.py tests/test_gpu_kv_attn.py tests/test_gpu_attn_merge_stress.py tests/test_gpu_host_runner.py tests/test_gpu_decoder.py tests/test_gpu_parity_depth.py -q -x --timeout 900 2>&1 | tail -12 | tee $OUT/pytest.log
for rep in 1 2 3; do
  r=$(timeout 300 python tools/attn_block_trace.py 2>&1 | grep "one launch" | sed 's/.*: *//; s/ us per.*//')
  echo "rep $rep -> $r" | tee -a $OUT/sweep.txt
done
make -C dash-infer_amd/csrc trace -j16 2>&1 | grep -E "error" | head
DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/attn_block_trace.py 2>&1 | tee $OUT/trace_7b.txt | tail -34
timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])" | tee -a $OUT/sweep.txt
timeout 300 python bench.py --workload tp8_rank_7b --no-extra --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tp8', d['value'], d['ms_per_step'])" | tee -a $OUT/sweep.txt
