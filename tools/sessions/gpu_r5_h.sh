#!/bin/bash
# round 5, call h: mid-round validation -- the whole GPU suite, then the default bench line exactly as the driver runs it
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5h
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -15 | tee gpurun_out/r5h/pytest_gpu.log
( time timeout 1500 python bench.py > gpurun_out/r5h/bench_default.json 2> gpurun_out/r5h/bench_default.err ) 2>&1 | tail -3 | tee gpurun_out/r5h/bench_time.txt
python - <<'PY' | tee gpurun_out/r5h/summary.txt
import json
d=json.loads([l for l in open("gpurun_out/r5h/bench_default.json") if l.startswith('{')][-1])
print("value", d["value"], d["unit"], "ms", d["ms_per_step"], "step frac", d["step_hbm"]["frac_of_peak"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","achieved_graph_events","frac_graph_events","avg_kernel_us_rocprof","avg_launch_us","traffic","rocprof_note")})
print("kernels", {k: v["avg_us"] for k, v in d["kernels"].items()})
print("cpu_baseline", d.get("cpu_baseline"))
for w in d.get("extra", {}).get("workloads", []):
    print(w.get("workload"), w.get("value"), w.get("unit"), w.get("ms_per_step"), (w.get("step_hbm") or {}).get("frac_of_peak"), w.get("error"), w.get("wall_s"))
print("top", [(k["kernel"][:60], k["calls"], k["avg_us"]) for k in d.get("rocprof_top_kernels", [])[:8]])
PY
