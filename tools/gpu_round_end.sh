#!/bin/bash
# One gpurun call for the end of a round: the GPU test suite, the bench line of every workload, the rocprofv3 kernel summary of
# the headline bench and the PMC traffic passes.  Everything lands under gpurun_out/$TAG; copy what is to be judged to profiles/.
#   gpurun --timeout 2400 -- 'TAG=r03z bash tools/gpu_round_end.sh'
TAG=${TAG:-rend}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -z "$SKIP_TESTS" ]; then
  DIHIP_FULL_DEPTH_ABLATION=${ABLATION:-0} timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "FULL DEPTH|configs\[|7B-width|operator graph|\[int4_b1\]|\[int4_b32_u4kv\]|\[cfg3_rank\]|passed|failed|error" | cut -c1-900 > $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
fi
timeout 900 python bench.py > $OUT/bench_int4_b1.json 2> $OUT/bench_int4_b1.err
python - <<PY
import json
d = json.load(open("$OUT/bench_int4_b1.json"))
print("int4_b1", d["value"], d["ms_per_step"], d["step_hbm"]["frac_of_peak"], d["roofline"]["frac"], {k: v["avg_us"] for k, v in d["kernels"].items()})
print("cpu_baseline", d.get("cpu_baseline"))
PY
for w in int8_b1 int4_b32_u4kv cfg3_rank moe_layer cfg5_moe prefill_2048; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$w.json"))
    print("$w", d["value"], d["ms_per_step"], d.get("step_hbm", {}).get("frac_of_peak"), {k: v["avg_us"] for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("$w FAILED", e)
PY
done
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o int4_b1 -- python $ROOT/bench.py --no-cpu-baseline --no-extra > $OUT/prof_bench.json 2> $OUT/prof_bench.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/bench_int4_b1_kernel_stats.csv && head -12 $OUT/bench_int4_b1_kernel_stats.csv | cut -c1-160
find $OUT/prof -name "*.csv" -size +4M -delete
cd $ROOT
if [ -z "$SKIP_PMC" ]; then
  for w in int4_b1 int4_b32_u4kv cfg3_rank int8_b1; do   # (counters for every workload, not the headline alone)
    bash tools/gpu_pmc.sh $TAG/pmc $w > $OUT/pmc_$w.log 2>&1
    tail -8 $OUT/pmc_$w.log | cut -c1-160
  done
fi
