#!/bin/bash
# round 3, call Q: packed-f32 fix-up (hipcc's SLP vectoriser) against scalar FMAs, with and without the ping-pong loop
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3q; mkdir -p $OUT; cd $ROOT
bench() {
  local name=$1; local w=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), {k: v["avg_us"] for k, v in d.get("gemms", {}).items()})
except Exception as e:
    print("bench $name FAILED", e)
PY
}
for v in pf_nopp pf_nopp_noslp pf_pp_noslp; do
  DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/$v timeout 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k prefill 2>&1 | tail -1 | cut -c1-300
  bench $v prefill_2048 DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/$v
done
