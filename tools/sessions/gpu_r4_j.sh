#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3; do
  for st in 64 20; do
    timeout 300 python bench.py --no-cpu-baseline --no-extra --runner host --steps $st --warmup 5 > gpurun_out/j_$i_$st.json 2>gpurun_out/j.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/j_$i_$st.json").read().strip().splitlines()[-1])
print("steps $st run $i: python", d["python_runner"], "host", d["host_runner"]["fused_graph"]["tokens_per_s"], d["host_runner"]["fused_graph"]["ms_per_step"])
PY
  done
done
