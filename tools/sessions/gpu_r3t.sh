#!/bin/bash
# round 3, call T: MoE block in 9 launches (route+group, finalize folded into the combine); down projection at M = 32 on the panel kernel
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3t; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_moe.py tests/test_gpu_host_ops.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x -k "moe or real_width or batched" 2>&1 | tail -3 | cut -c1-400
timeout 300 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_kv_attn.py -m gpu -q -x 2>&1 | tail -2 | cut -c1-400
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
bench moe_fused cfg5_moe X=1
bench moe_sep cfg5_moe DIHIP_MOE_FUSED=0
bench b32 int4_b32_u4kv X=1
