#!/bin/bash
# round 5, call z: does the MLP block (gate/up + SwiGLU -> down in one launch) pay at the per-rank shapes of TP = 8 (intermediate 2432: a 10 KB hand-off)?
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5z
for m in 0 1; do
  DIHIP_DECODER_MLP_BLOCK=$m timeout 300 python bench.py --workload tp8_rank_7b --runner python --no-cpu-baseline --no-extra > gpurun_out/r5z/tp8_mlp$m.json 2> gpurun_out/r5z/tp8_mlp$m.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r5z/tp8_mlp$m.json"))
    print("tp8_rank_7b mlp_block=$m", d["value"], d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items()})
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/r5z/tp8_mlp$m.err").read()[-600:])
PY
done
