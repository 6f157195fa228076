#!/bin/bash
# round 3, call P: SQ counters of the prefill SwiGLU GEMM (one PMC pass, kernel trace only)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3p; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for lib in default pf_nopp; do
  if [ $lib = default ]; then unset DIHIP_LIB_DIR; else export DIHIP_LIB_DIR=$ROOT/dash-infer_amd/lib/$lib; fi
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/$lib -o pmc -- python $ROOT/tools/pf_gemm_probe.py > $OUT/$lib.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/${lib}_b -o pmc -- python $ROOT/tools/pf_gemm_probe.py > $OUT/${lib}_b.log 2>&1
  python - <<PY
import csv, glob, collections
for d in ("$OUT/$lib", "$OUT/${lib}_b"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_prefill" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("$lib", {k: round(sum(v[1:]) / max(1, len(v) - 1)) for k, v in acc.items()})
PY
done
find $OUT -name "*.csv" -size +2M -delete
