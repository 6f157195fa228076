#!/bin/bash
# r6 z3: smoke() + the driver's command once more after the line's runner-up object was renamed (bench.py only; kernels unchanged)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6z3
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.log
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default_time.txt
python - <<PY
import json
lines = [l for l in open("$OUT/bench_default.json").read().splitlines() if l.strip()]
print("stdout lines:", len(lines), "bytes of the last:", len(lines[-1]))
d = json.loads(lines[-1])
print(d["value"], d["ms_per_step"], [k for k in d if k.startswith("roofline")], d["roofline"]["kernel"][:50], d["roofline"]["frac"])
r2 = d.get("roofline_attn_block") or d.get("roofline_gemv")
print(r2["kernel"][:50], r2["frac"], r2["traffic"], r2["avg_kernel_us_rocprof"])
PY
tail -3 $OUT/bench_default_time.txt
