#!/bin/bash
# round 5, call y: the default bench line again (driver-style) after the last bench.py change (prefill through the C++ layer; PMC summary of the tree committed)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5y
( time timeout 900 python bench.py > gpurun_out/r5y/bench_default.json 2> gpurun_out/r5y/bench_default.err ) 2> gpurun_out/r5y/bench_default_time.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5y/bench_default.json"))
print("int4_b1", d["value"], d["ms_per_step"], d["step_hbm"]["frac_of_peak"], d["roofline"]["frac"], d["roofline"]["traffic"])
for w in d.get("extra", {}).get("workloads", []):
    print(" ", w.get("workload"), w.get("value"), w.get("ms_per_step"), (w.get("step_hbm") or {}).get("frac_of_peak"), (w.get("roofline") or {}).get("frac"), (w.get("roofline") or {}).get("traffic"))
PY
tail -3 gpurun_out/r5y/bench_default_time.txt
