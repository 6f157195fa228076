#!/bin/bash
# round 3, call S: small-batch kernel choice for the small matrices (qkv / o) of configs[2] and configs[3]
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3s; mkdir -p $OUT; cd $ROOT
bench() {
  local name=$1; local w=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("bench $name FAILED", e)
PY
}
bench cfg3_base cfg3_rank X=1
bench cfg3_ksl2 cfg3_rank DIHIP_GEMM_KSLICE=2
bench cfg3_ksl0 cfg3_rank DIHIP_GEMM_KSLICE=0
bench b32_ksl2 int4_b32_u4kv DIHIP_GEMM_KSLICE=2
bench b32_ksl0 int4_b32_u4kv DIHIP_GEMM_KSLICE=0
