#!/bin/bash
# round 4: small-batch knob sweep (cfg3_rank, int4_b32_u4kv)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, workload, env...
  local name=$1; local wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --no-extra --no-cpu-baseline --steps 20 --warmup 4 --blocks 3 > gpurun_out/c_$name.json 2> gpurun_out/c_$name.err
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/c_{n}.json").read().strip().splitlines()[-1])
    k=d.get("kernels",{})
    print(n, d["value"], d["ms_per_step"], d["step_hbm"]["frac_of_peak"], {a:b["avg_us"] for a,b in k.items() if a!="lm_head"}, flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/c_{n}.err").read()[-300:])
PY
}
run cfg3_base cfg3_rank A=1
run cfg3_ksl2 cfg3_rank DIHIP_GEMM_KSLICE=2
run cfg3_ksl0 cfg3_rank DIHIP_GEMM_KSLICE=0
run cfg3_panel25 cfg3_rank DIHIP_PANEL_ONE_SLICE_PCT=25
run cfg3_wgs2 cfg3_rank DIHIP_ATTN_WGS_PER_CU=2
run b32_base int4_b32_u4kv A=1
run b32_wgs2 int4_b32_u4kv DIHIP_ATTN_WGS_PER_CU=2
run b32_nonf int4_b32_u4kv DIHIP_DECODER_NORM_FUSE=0
run b32_merge_launch int4_b32_u4kv DIHIP_DECODER_ATTN_MERGE=launch
