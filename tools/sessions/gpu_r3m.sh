#!/bin/bash
# round 3, call M: prefill GEMM wave-grid / row-sum variants (lib/pf_*), and the split count of the u4 attention at batch 32
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3m; mkdir -p $OUT; cd $ROOT
bench() {
  local name=$1; local w=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), {k: v["avg_us"] for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("bench $name FAILED", e)
PY
}
bench pf_base prefill_2048 X=1
for v in wm2 ms wm2ms; do
  DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/pf_$v timeout 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k prefill 2>&1 | tail -2 | cut -c1-300
  bench pf_$v prefill_2048 DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/pf_$v
done
bench b32_s2 int4_b32_u4kv DIHIP_ATTN_NSPLITS=2
bench b32_s8 int4_b32_u4kv DIHIP_ATTN_NSPLITS=8
bench b32_s16 int4_b32_u4kv DIHIP_ATTN_NSPLITS=16
