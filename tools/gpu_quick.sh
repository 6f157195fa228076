#!/bin/bash
# quick loop: selected tests + bench (+ optional rocprof) 
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$TESTS" ]; then timeout 900 python -m pytest $TESTS -q -m gpu -x --timeout 600 > $OUT/tests.log 2>&1; echo "tests exit $?"; tail -4 $OUT/tests.log; fi
timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("tok/s", d["value"], "ms/step", d["ms_per_step"], "frac", d["step_hbm"]["frac_of_peak"])
for k,v in d.get("kernels",{}).items(): print("  %-28s %8.2f us  %7.1f GB/s" % (k, v["avg_us"], v["GBps"]))
PY
if [ "${PROFILE:-0}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline ${BENCH_ARGS:-} > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err)
  grep dihip $OUT/prof/bench_kernel_stats.csv | grep -v pack_ | cut -d, -f1-4 | cut -c1-150
  find $OUT/prof -name "*kernel_trace*" -size +8M -delete
fi
