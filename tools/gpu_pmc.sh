#!/bin/bash
# HBM traffic of the hot kernels: separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over a short bench run
#   bash tools/gpu_pmc.sh <tag> [workload]      workload: a bench.py --workload name (default int4_b1); the summary of any other
#   workload is written as pmc_hbm_traffic_<workload>.csv (bench.py's pmc_traffic() picks the file of its workload)
TAG=${1:-pmc}
WL=${2:-int4_b1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${WL}_$ctr -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 4 --warmup 2 --blocks 1 --no-cpu-baseline --no-graph --runner python --no-extra > $OUT/${WL}_$ctr.json 2> $OUT/${WL}_$ctr.err
  echo "$WL $ctr exit $?"
  ls $OUT/${WL}_$ctr | head -3
done
HASH=$(cd $GRAFT_REPO_ROOT && python -c "import bench; print(bench.csrc_tree_hash())")
python - <<PY
# per-kernel averages -> $OUT/pmc_hbm_traffic.csv (the schema bench.py's pmc_traffic() reads from profiles/)
import csv, glob, collections
NOTE = {"FETCH_SIZE": "FETCH_SIZE x2: gfx950 reports half the bytes of wide coalesced streams (MI355X_MICROARCH.md, HBM)",
        "WRITE_SIZE": "WRITE_SIZE as reported"}
rows = []
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("$OUT/${WL}_%s/*counter_collection*.csv" % ctr)
    if not files: print("no counter file for", ctr); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if "dihip" not in r["Kernel_Name"] or "pack" in r["Kernel_Name"]: continue
            a = acc[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        kb = v / n   # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
        rows.append((ctr, k, n, round(kb, 1), int(kb * 1024 * (2 if ctr == "FETCH_SIZE" else 1)), NOTE[ctr], "$HASH"))
        print("%-10s %-80s n=%5d  avg %.1f KB" % (ctr, k[:80], n, kb))
suffix = "" if "$WL" == "int4_b1" else "_$WL"
with open("$OUT/pmc_hbm_traffic%s.csv" % suffix, "w", newline="") as f:
    w = csv.writer(f); w.writerow(["counter", "kernel", "dispatches", "avg_counter_KB_raw", "avg_bytes_corrected", "note", "csrc_hash"]); w.writerows(rows)
PY
find $OUT -name "*.csv" -size +8M -delete
