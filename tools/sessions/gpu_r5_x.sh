#!/bin/bash
# round 5, call x: the context phase of the prefill workload through the C++ operator layer
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5x
timeout 600 python bench.py --workload prefill_2048 --no-cpu-baseline --no-extra > gpurun_out/r5x/bench_prefill.json 2> gpurun_out/r5x/bench_prefill.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5x/bench_prefill.json"))
print(d["value"], d["ms_per_step"], d["runner"]); print("python", d["python_runner"]); print("host", d["host_runner"])
PY
tail -3 gpurun_out/r5x/bench_prefill.err
