#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --runner python --no-extra --no-cpu-baseline --steps 30 --warmup 5 --blocks 3 > gpurun_out/e_$name.json 2> gpurun_out/e_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/e_{n}.json").read().strip().splitlines()[-1])
    k=d.get("kernels",{})
    print(n, d["value"], d["ms_per_step"], {a:b["avg_us"] for a,b in k.items() if a!="lm_head"}, flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/e_{n}.err").read()[-300:])
PY
}
run base A=1
run nomerge DIHIP_ATTN_MERGE=none
run launchmerge DIHIP_ATTN_MERGE=launch
timeout 600 python -m pytest tests/test_gpu_host_ops.py -x -q -k "preprocess" 2>&1 | tail -3
