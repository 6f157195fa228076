#!/bin/bash
# round 3, call N: scheduling variants of the prefill GEMM's multiply loop (lib/pf_s1..3) against the default (MFMA row sums on)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3n; mkdir -p $OUT; cd $ROOT
bench() {
  local name=$1; local w=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), d.get("gemms"), d.get("attention", {}).get("tflops"))
except Exception as e:
    print("bench $name FAILED", e)
PY
}
timeout 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k prefill 2>&1 | tail -2 | cut -c1-300
bench pf_base prefill_2048 X=1
for v in s1 s2 s3; do
  DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/pf_$v timeout 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k prefill 2>&1 | tail -2 | cut -c1-300
  bench pf_$v prefill_2048 DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/pf_$v
done
