#!/bin/bash
# round 3, call Z: per-kernel times of configs[3] (one TP = 8 rank) and configs[2] (batch 32)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3z; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for w in cfg3_rank int4_b32_u4kv; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o k -- python $ROOT/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${w}_kernel_stats.csv
  python - <<PY
import csv
print("$w")
for r in csv.DictReader(open("$OUT/${w}_kernel_stats.csv")):
    if "dihip" in r["Name"] and int(r["Calls"]) >= 100 and "pack" not in r["Name"]:
        print("  %-110s %6s %8.2f" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1000))
PY
done
find $OUT -name "*.csv" -size +4M -delete
