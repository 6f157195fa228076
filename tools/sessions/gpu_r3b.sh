#!/bin/bash
# round 3, call B: parity of the changed kernels, full-depth ablation table, A/B of the ring fill and the attention merge
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3b; mkdir -p $OUT; cd $ROOT
( time timeout 600 python -m pytest tests/test_gpu_kv_attn.py tests/test_gpu_gemm.py tests/test_gpu_glue.py -m gpu -q -x ) > $OUT/pytest_kernels.log 2>&1
tail -4 $OUT/pytest_kernels.log
( time timeout 300 python -m pytest tests/test_gpu_decoder.py -m gpu -q -s -k "greedy_decode_matches_oracle or real_width or qwen7b_width" ) > $OUT/pytest_decoder.log 2>&1
tail -4 $OUT/pytest_decoder.log
bench() {  # name, env...
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
bench default X=1
bench merge_launch DIHIP_DECODER_ATTN_MERGE=launch
bench merge_launch_dyn_tps DIHIP_DECODER_ATTN_MERGE=launch DIHIP_ATTN_STATIC_TPS=0
bench ring8 DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/ring8
bench ring2 DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/ring2
bench ring6 DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/ring6
( time DIHIP_FULL_DEPTH_ABLATION=1 timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -q -s -k "full_depth" ) > $OUT/pytest_fulldepth_ablation.log 2>&1
tail -4 $OUT/pytest_fulldepth_ablation.log
