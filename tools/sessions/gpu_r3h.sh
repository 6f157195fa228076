#!/bin/bash
# round 3, call H: bench lines of the secondary workloads after the decode-step changes (profiles/r03h_*)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3h; mkdir -p $OUT; cd $ROOT
for w in int4_b32_u4kv cfg3_rank int8_b1 cfg5_moe; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$w.json"))
    print("$w", d["value"], d["ms_per_step"], d.get("step_hbm", {}).get("frac_of_peak"), {k: v["avg_us"] for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("$w FAILED", e)
PY
done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b32 -o b32 -- python $ROOT/bench.py --workload int4_b32_u4kv --no-cpu-baseline --steps 16 > $OUT/prof_b32.json 2> $OUT/prof_b32.err
f=$(find $OUT/prof_b32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_int4_b32_u4kv_kernel_stats.csv && head -16 $OUT/bench_int4_b32_u4kv_kernel_stats.csv | cut -c1-200
find $OUT/prof_b32 -name "*.csv" -size +2M -delete
