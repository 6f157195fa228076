#!/bin/bash
# round 3, call O: the ping-pong main loop of the prefill GEMM (default build) against the lockstep loop (lib/pf_nopp)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3o; mkdir -p $OUT; cd $ROOT
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
timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k prefill 2>&1 | tail -5 | cut -c1-600
bench pf_pp prefill_2048 X=1
bench pf_nopp prefill_2048 DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/pf_nopp
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x -k "greedy or prefill or context" 2>&1 | tail -3 | cut -c1-600
