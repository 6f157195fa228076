#!/bin/bash
# HBM traffic of the hot kernels: separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over a short bench run
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$ctr -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > $OUT/$ctr.json 2> $OUT/$ctr.err
  echo "$ctr exit $?"
  ls $OUT/$ctr | head
done
python - <<PY
import csv, glob, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("$OUT/%s/*counter_collection*.csv" % ctr)
    if not files: print("no counter file for", ctr); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(files[0]) as f:
        rd = csv.DictReader(f)
        for r in rd:
            if "dihip" not in r["Kernel_Name"] or "pack" in r["Kernel_Name"]: continue
            a = acc[r["Kernel_Name"][:80]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-10s %-80s n=%5d  avg %.1f KB" % (ctr, k, n, v / n))
PY
find $OUT -name "*.csv" -size +8M -delete
