#!/bin/bash
# One gpurun call for the end of a round: the GPU test suite, the default bench line (headline + extra.workloads), the workloads
# that are not in it, rocprofv3 kernel summaries and the PMC traffic passes.  Everything lands under gpurun_out/$TAG; copy what is
# to be judged to profiles/.
#   gpurun --timeout 3000 -- 'TAG=r05z bash tools/gpu_round_end.sh'
TAG=${TAG:-rend}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -z "$SKIP_TESTS" ]; then
  DIHIP_FULL_DEPTH_ABLATION=${ABLATION:-0} timeout 2400 python -m pytest tests -m gpu -q -s --timeout 900 2>&1 | grep -E "FULL DEPTH|configs\[|7B-width|operator graph|\[int4_b1\]|\[int4_b32_u4kv\]|\[cfg3_rank\]|\[kv codec|\[deferred norm|\[prefill tail|passed|failed|error" | cut -c1-900 > $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
fi
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default_time.txt
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
python - <<PY
import json
lines = [l for l in open("$OUT/bench_default.json").read().splitlines() if l.strip()]
print("stdout lines:", len(lines), "bytes of the last:", len(lines[-1]))
d = json.loads(lines[-1])
print("int4_b1", d["value"], d["ms_per_step"], d["step_hbm"]["frac_of_peak"], "roofline", d["roofline"]["kernel"][:40], d["roofline"]["frac"], d["roofline"].get("traffic"),
      "runner-up", (d.get("roofline_attn_block") or d.get("roofline_gemv") or {}).get("kernel", "")[:40], (d.get("roofline_attn_block") or d.get("roofline_gemv") or {}).get("frac"),
      (d.get("roofline_attn_block") or d.get("roofline_gemv") or {}).get("traffic"), d.get("kernels_us"))
print("python runner", d.get("python_runner_tokens_per_s"), "cpu_baseline", (d.get("cpu_baseline") or {}).get("value"), "wall_s", d.get("wall_s"))
for w in d.get("extra", []):
    print(" ", w)
PY
tail -3 $OUT/bench_default_time.txt
for w in moe_layer; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2> $OUT/bench_$w.err | tail -1 > $OUT/bench_$w.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$w.json"))
    print("$w", d["value"], d["ms_per_step"], d.get("step_hbm", {}).get("frac_of_peak"))
except Exception as e:
    print("$w FAILED", e)
PY
done
export TMPDIR=/tmp
cd /tmp
for w in int4_b1 int8_b1 int4_b32_u4kv cfg3_rank tp8_rank_7b prefill_2048; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python $ROOT/bench.py --workload $w --no-cpu-baseline --no-extra > $OUT/prof_bench_$w.json 2> $OUT/prof_bench_$w.err
  f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/bench_${w}_kernel_stats.csv && echo "== $w" && grep dihip $OUT/bench_${w}_kernel_stats.csv | head -8 | cut -d, -f1-4 | cut -c1-170
  find $OUT/prof_$w -name "*.csv" -size +4M -delete
done
cd $ROOT
if [ -z "$SKIP_PMC" ]; then
  for w in int4_b1 int4_b32_u4kv cfg3_rank int8_b1; do   # (counters for every workload, not the headline alone)
    bash tools/gpu_pmc.sh $TAG/pmc $w > $OUT/pmc_$w.log 2>&1
    tail -8 $OUT/pmc_$w.log | cut -c1-160
  done
fi
