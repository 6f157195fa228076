#!/bin/bash
# round 4, call ah: L2 counters of the K-slice GEMM at M = 32 (7B gate/up pair) with the activation fragments loaded nontemporal
# (lib/ksl_nt: the form before round 4) and temporal (product): one PMC pass each, counters only
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r4ah
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in ksl_nt product; do
  L=$R/dash-infer_amd/lib; [ $v = ksl_nt ] && L=$R/dash-infer_amd/lib/ksl_nt
  SHAPE=gate_up_swiglu LD_LIBRARY_PATH=$L timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/r4ah/$v -o pmc -- $R/tools/gemv_bench 4 128 32 2 > $R/gpurun_out/r4ah/$v.log 2>&1
  echo "$v exit $?"
done
python - <<PY
import csv, glob, collections
for v in ("ksl_nt", "product"):
    files = glob.glob("$R/gpurun_out/r4ah/%s/*counter_collection*.csv" % v)
    if not files: print(v, "no counter file"); continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if "gemm_kslice_kernel" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    row = {k: v_[1] / v_[0] for k, v_ in acc.items()}
    n = next(iter(acc.values()))[0] if acc else 0
    hit = row.get("TCC_HIT_sum", 0) / max(1.0, row.get("TCC_HIT_sum", 0) + row.get("TCC_MISS_sum", 0))
    print("%-8s launches %4d  " % (v, n) + "  ".join("%s %.4g" % kv for kv in sorted(row.items())) + "  L2 hit rate %.3f" % hit)
PY
find $R/gpurun_out/r4ah -name "*.csv" -size +4M -delete
