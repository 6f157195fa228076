#!/bin/bash
# round 3, call C: timelines of the decode GEMV (new ring fill) and the decode-step attention, the new operator-graph / TP tests, runtime knobs
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3c; mkdir -p $OUT; cd $ROOT
( time timeout 600 python -m pytest tests/test_gpu_host_graph.py tests/test_gpu_host_ops.py tests/test_gpu_tp_loopback.py tests/test_gpu_p2p_processes.py -m gpu -q -x -s ) > $OUT/pytest_host_tp.log 2>&1
tail -6 $OUT/pytest_host_tp.log
export LD_LIBRARY_PATH=$ROOT/dash-infer_amd/lib/trace:/opt/rocm/lib
TRACE=1 TRACE_BINS=1 timeout 300 ./tools/gemv_bench 4 128 1 > $OUT/gemv_trace.log 2>&1
grep -v "warm\|pre-trace\|trace launch" $OUT/gemv_trace.log | head -150
timeout 120 ./tools/attn_bench 1 2048 0 > $OUT/attn_trace_inlaunch.log 2>&1; cat $OUT/attn_trace_inlaunch.log
MERGE=launch timeout 120 ./tools/attn_bench 1 2048 0 > $OUT/attn_trace_launch.log 2>&1; cat $OUT/attn_trace_launch.log
unset LD_LIBRARY_PATH
bench() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --layers 8 --steps 48 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("$name (8 layers)", d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
except Exception as e:
    print("bench $name FAILED", e)
PY
}
bench base X=1
bench optflush0 AMD_OPT_FLUSH=0
bench optflush1 AMD_OPT_FLUSH=1
bench pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
bench pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
