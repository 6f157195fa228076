#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3f; mkdir -p $OUT; cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_host_graph.py tests/test_gpu_kv_attn.py tests/test_gpu_gemv_stress.py -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log | cut -c1-400
bench() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
except Exception as e:
    print("bench $name FAILED", e)
PY
}
bench designated X=1
bench ticket DIHIP_ATTN_MERGE=ticket
export LD_LIBRARY_PATH=$ROOT/dash-infer_amd/lib/trace:/opt/rocm/lib
timeout 120 ./tools/attn_bench 1 2048 0 > $OUT/attn_trace_designated.log 2>&1; cat $OUT/attn_trace_designated.log
unset LD_LIBRARY_PATH
timeout 300 python bench.py --workload prefill_2048 --steps 8 --warmup 2 > $OUT/bench_prefill_2048.json 2> $OUT/bench_prefill.err; cat $OUT/bench_prefill_2048.json | cut -c1-1500; tail -3 $OUT/bench_prefill.err
