#!/bin/bash
# round 4, call af: what-if builds of the context-phase GEMM (pfxN: -DDIHIP_PF_X=N; 1 no fix-up, 2 no A staging, 4 no barrier,
# 8 no nibble expansion, 15 all four): the four GEMMs' times inside the prefill of one 2048-token prompt
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for v in "" pfx1 pfx2 pfx4 pfx8 pfx15; do
  DIHIP_LIB_DIR=${v:+$GRAFT_REPO_ROOT/dash-infer_amd/lib/$v} timeout 300 python bench.py --workload prefill_2048 --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
g=d['gemms']
print('%-8s' % '${v:-product}', 'ms', d['ms_per_step'], {k: (v['avg_us'], v['tflops']) for k, v in g.items()})
"
done 2>&1 | tee gpurun_out/r4af_prefill_gemm_whatif.txt
