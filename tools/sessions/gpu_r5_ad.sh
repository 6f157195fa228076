#!/bin/bash
# round 5, call ad: last sanity pass on the final tree: smoke(), the headline quickly, the end-to-end decoder / host-runner tests
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5ad
{
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python bench.py --steps 32 --blocks 3 --no-cpu-baseline --no-extra > gpurun_out/r5ad/bench_quick.json 2> gpurun_out/r5ad/bench_quick.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5ad/bench_quick.json"))
print("int4_b1", d["value"], d["ms_per_step"], "python", d["python_runner"]["tokens_per_s"], "traffic", d["roofline"]["traffic"], d["roofline"]["frac"])
PY
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_host_runner.py tests/test_gpu_attn_block.py -q -m gpu --timeout 600 -x 2>&1 | tail -3
} 2>&1 | tee gpurun_out/r5ad/log.txt
