#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3l; mkdir -p $OUT; cd $ROOT
( time timeout 900 python -m pytest tests/test_gpu_kv_attn.py tests/test_gpu_host_ops.py tests/test_gpu_tp_loopback.py tests/test_gpu_moe.py -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log | cut -c1-400
( time timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x -k "greedy_decode or real_width or moe" ) > $OUT/pytest_dec.log 2>&1
tail -5 $OUT/pytest_dec.log | cut -c1-400
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
bench b32_inlaunch int4_b32_u4kv X=1
bench b32_launch int4_b32_u4kv DIHIP_ATTN_MERGE=launch
bench cfg3_inlaunch cfg3_rank X=1
bench cfg3_launch cfg3_rank DIHIP_ATTN_MERGE=launch
