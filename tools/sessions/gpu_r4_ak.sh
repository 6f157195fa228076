#!/bin/bash
# round 4, call ak: the decode GEMV's weight stream without the nt bit (lib/gemv_nont) against the product build: headline step
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for rep in 1 2; do for v in product gemv_nont; do
  D=""; [ $v = gemv_nont ] && D=$GRAFT_REPO_ROOT/dash-infer_amd/lib/gemv_nont
  DIHIP_LIB_DIR=$D timeout 200 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-extra --runner python 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-10s' % '$v', 'tok/s', d['value'], 'ms', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items()})
"
done; done 2>&1 | tee gpurun_out/r4ak_gemv_nont.txt
