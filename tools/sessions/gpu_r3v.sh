#!/bin/bash
# round 3, call V: same-box A/B of the MoE step (fused / separate) and the headline, twice
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3v; mkdir -p $OUT; cd $ROOT
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
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4
bench b1 int4_b1 X=1
bench moe_fused cfg5_moe X=1
bench moe_sep cfg5_moe DIHIP_MOE_FUSED=0
bench moe_fused2 cfg5_moe X=1
bench b32 int4_b32_u4kv X=1
bench b1_again int4_b1 X=1
