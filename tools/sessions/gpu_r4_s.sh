#!/bin/bash
# round 4, call s: prefill GEMM with temporal weight loads (lib/pfw: -DDIHIP_PF_TEMPORAL_W) against the product build:
# bench line + HBM fetch per launch
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
{
for v in "" pfw ""  pfw; do
  DIHIP_LIB_DIR=${v:+$GRAFT_REPO_ROOT/dash-infer_amd/lib/$v} timeout 300 python bench.py --workload prefill_2048 --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('${v:-base}', d['value'], d['unit'], 'ms', d['ms_per_step'], json.dumps(d.get('gemms'))[:400])
"
done
} 2>&1 | tee gpurun_out/r4s_prefill_temporal_w.txt
export DIHIP_LIB_DIR=$GRAFT_REPO_ROOT/dash-infer_amd/lib/pfw
bash tools/gpu_pmc.sh r4pfw prefill_2048 2>&1 | grep "gemm_prefill" | tee -a gpurun_out/r4s_prefill_temporal_w.txt
