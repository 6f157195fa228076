#!/bin/bash
# r6 k: runner tests after the TP-tail change + the bench's N > 1 C++ leg as a one-rank plumbing check
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_host_runner.py tests/test_gpu_sampling.py tests/test_gpu_host_graph.py -q --timeout 900 2>&1 | tail -8 | tee $OUT/pytest.log
DIHIP_BENCH_HOST_TP=force timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 16 --warmup 4 > $OUT/bench_tp_selftest.out 2> $OUT/bench_tp_selftest.err
tail -c 1500 $OUT/bench_tp_selftest.out; tail -5 $OUT/bench_tp_selftest.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_detail.json"))
print(json.dumps(d["host_runner"].get("tp_leg_selftest_one_rank"), indent=1)[:1500])
print(d["host_runner"]["fused_graph"]["tokens_per_s"])
PY
