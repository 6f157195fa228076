#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3e; mkdir -p $OUT; cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_host_graph.py tests/test_gpu_kv_attn.py tests/test_gpu_gemm.py tests/test_gpu_glue.py tests/test_gpu_moe.py -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_default.json"))
    print("default", d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
except Exception as e:
    print("bench FAILED", e)
PY
export LD_LIBRARY_PATH=$ROOT/dash-infer_amd/lib/trace:/opt/rocm/lib
timeout 120 ./tools/attn_bench 1 2048 0 > $OUT/attn_trace_inlaunch.log 2>&1; cat $OUT/attn_trace_inlaunch.log
TRACE=1 timeout 300 ./tools/gemv_bench 4 128 1 > $OUT/gemv_trace.log 2>&1
grep -A9 "plan:" $OUT/gemv_trace.log | grep -v "warm\|pre-trace\|trace launch" | head -60
