#!/bin/bash
# round 3, call A: the new full-depth parity tests (timed), and the headline bench with kernel arguments in host vs device memory
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3a; mkdir -p $OUT; cd $ROOT
nproc > $OUT/host.txt; free -g >> $OUT/host.txt
( time timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -q -s -k "full_depth or real_width" ) > $OUT/pytest_fulldepth.log 2>&1
tail -5 $OUT/pytest_fulldepth.log
for kv in 0 1; do
  HIP_FORCE_DEV_KERNARG=$kv timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_devkernarg$kv.json 2> $OUT/bench_devkernarg$kv.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_devkernarg$kv.json"))
    print("HIP_FORCE_DEV_KERNARG=$kv", d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
except Exception as e:
    print("bench $kv FAILED", e)
PY
done
