#!/bin/bash
# round 4, batch-1 knob sweep: K-slice skew / wave mapping of the decode GEMV, attention split count
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 5 --blocks 3 > gpurun_out/b_$name.json 2> gpurun_out/b_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/b_{n}.json").read().strip().splitlines()[-1])
    k=d.get("kernels",{})
    print(n, d["value"], d["ms_per_step"], {a:b["avg_us"] for a,b in k.items() if a!="lm_head"}, d.get("last_ids"), flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/b_{n}.err").read()[-300:])
PY
}
run base A=1
run skew6 DIHIP_GEMV_KSKEW=6
run skew10 DIHIP_GEMV_KSKEW=10
run skew14 DIHIP_GEMV_KSKEW=14
run wmap DIHIP_GEMV_WMAP=1
run wmap_skew6 DIHIP_GEMV_WMAP=1 DIHIP_GEMV_KSKEW=6
run wmap_skew10 DIHIP_GEMV_WMAP=1 DIHIP_GEMV_KSKEW=10
run wmap_skew14 DIHIP_GEMV_WMAP=1 DIHIP_GEMV_KSKEW=14
run splits24 DIHIP_ATTN_NSPLITS=24
run splits34 DIHIP_ATTN_NSPLITS=34
run base2 A=1
