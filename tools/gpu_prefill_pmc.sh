#!/bin/bash
# SQ counters of the prefill attention kernel at one length (two PMC passes; counters only, no tracing domains)
TAG=${1:-pfpmc}
L=${2:-8192}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py $L > $OUT/p$i.log 2>&1
  echo "pass $i exit $?"
done
python - <<PY
import csv, glob, collections
for i in (1, 2):
    files = glob.glob("$OUT/p%d/*counter_collection*.csv" % i)
    if not files: print("no counter file for pass", i); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if "prefill_attn" not in r["Kernel_Name"]: continue
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(acc.items()):
        print("%-28s n=%3d  avg %.4g" % (k, n, v / n))
PY
find $OUT -name "*.csv" -size +8M -delete
