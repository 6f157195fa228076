#!/bin/bash
# quick session: selected tests + bench A/B via env
OUT=gpurun_out/${TAG:-r02b}
mkdir -p $OUT
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$PWD/dash-infer_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest $TESTS -q -m gpu -x --timeout 600 > $OUT/tests.log 2>&1; echo "tests exit $?"; tail -6 $OUT/tests.log; fi
run_bench() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name exit $?"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    print("$name: tok/s", d["value"], "ms/step", d["ms_per_step"], "frac", d["step_hbm"]["frac_of_peak"])
    for k,v in d.get("kernels",{}).items(): print("  %-28s %8.2f us  %7.1f GB/s" % (k, v["avg_us"], v["GBps"]))
except Exception as e:
    print("$name: no result", e); print(open("$OUT/bench_$name.err").read()[-1500:])
PY
}
for v in ${VARIANTS:-base}; do
  case $v in
    base) run_bench base X=1 ;;
    nomerge) run_bench nomerge DIHIP_ATTN_MERGE_IN_OPROJ=0 ;;
    *) run_bench $v $(echo $v | tr ',' ' ') ;;
  esac
done

if [ "${PROFILE:-0}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline ${BENCH_ARGS:-} > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err)
  grep dihip $OUT/prof/bench_kernel_stats.csv | grep -v pack_ | cut -d, -f1-4 | cut -c1-200
  find $OUT/prof -name "*kernel_trace*" -size +8M -delete
fi
