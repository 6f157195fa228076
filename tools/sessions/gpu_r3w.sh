#!/bin/bash
# round 3, call W: split count of skinny dense GEMMs (MoE router / gate)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3w; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_moe.py tests/test_gpu_gemm.py tests/test_gpu_host_graph.py -m gpu -q -x 2>&1 | tail -2 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x -k "moe" 2>&1 | tail -2 | cut -c1-300
export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o moe -- python $ROOT/bench.py --workload cfg5_moe --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/moe_kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/moe_kernel_stats.csv")):
    if "dihip" in r["Name"] and int(r["Calls"]) >= 100 and "pack" not in r["Name"]:
        print("  %-100s %6s %8.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1000))
PY
find $OUT -name "*.csv" -size +4M -delete
cd $ROOT
timeout 300 python bench.py --workload cfg5_moe --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; python -c "
import json; d=json.load(open('$OUT/bench_cfg5.json')); print('cfg5_moe', d['value'], d['ms_per_step'])"
