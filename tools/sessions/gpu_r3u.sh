#!/bin/bash
# round 3, call U: per-kernel times of the MoE decode step, fused (9 launches per block) against separate (11)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3u; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for m in 1 0; do
  DIHIP_MOE_FUSED=$m timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$m -o moe -- python $ROOT/bench.py --workload cfg5_moe --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench$m.json 2> $OUT/bench$m.err
  f=$(find $OUT/prof$m -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/moe_fused${m}_kernel_stats.csv
  python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/moe_fused${m}_kernel_stats.csv")))
print("DIHIP_MOE_FUSED=$m")
for r in rows:
    if "dihip" in r["Name"] and int(r["Calls"]) >= 100:
        print("  %-110s %6s %8.2f" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1000))
PY
done
find $OUT -name "*.csv" -size +4M -delete
