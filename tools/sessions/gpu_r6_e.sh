#!/bin/bash
# r6 e: attention block: ring 4+4, per-wave o sweep, staggered polls -- tests, timeline, A/B of each against the build with all three
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_attn_block.py tests/test_gpu_attn_merge_stress.py -q -x --timeout 600 2>&1 | tail -8 | tee $OUT/pytest.log
DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/trace timeout 300 python tools/attn_block_trace.py 2>&1 | tee $OUT/trace_7b.txt
for rep in 1 2; do
for V in "" ab_early8 ab_nosweep ab_nostag; do
  D=$PWD/dash-infer_amd/lib${V:+/$V}
  echo "== variant ${V:-product} (rep $rep)" | tee -a $OUT/ab.txt
  DIHIP_LIB_DIR=$D timeout 300 python tools/attn_block_trace.py 2>&1 | grep "us per layer" | tee -a $OUT/ab.txt
done
done
for V in "" ab_early8 ab_nosweep ab_nostag; do
  D=$PWD/dash-infer_amd/lib${V:+/$V}
  DIHIP_LIB_DIR=$D timeout 300 python bench.py --no-extra --no-cpu-baseline --runner python --steps 32 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${V:-product}', d['value'], d['ms_per_step'], d.get('kernels_us'))" | tee -a $OUT/bench_ab.txt
done
