#!/bin/bash
# r6 d: polled split records in the attention block: tests + timeline A/B
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_attn_block.py tests/test_gpu_attn_merge_stress.py -q -x --timeout 600 2>&1 | tail -8 | tee $OUT/pytest.log
for P in 0 1; do
  echo "== DIHIP_ATTN_BLOCK_POLLED=$P" | tee -a $OUT/trace_7b.txt
  DIHIP_ATTN_BLOCK_POLLED=$P DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/attn_block_trace.py 2>&1 | tee -a $OUT/trace_7b.txt
  DIHIP_ATTN_BLOCK_POLLED=$P timeout 300 python tools/attn_block_trace.py 2>&1 | grep "us per layer" | tee -a $OUT/trace_7b.txt
done
for P in 0 1; do
  DIHIP_ATTN_BLOCK_POLLED=$P timeout 300 python bench.py --no-extra --no-cpu-baseline --runner python --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('POLLED=$P', d['value'], d['ms_per_step'], d.get('kernels_us'))" | tee -a $OUT/bench_ab.txt
done
