#!/bin/bash
# r6 f: attention block: sleeps before the polls, sentinel spreading -- sweep on the graph-timed 8-layer tool
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() {  # run <label> <libdir-suffix> env...
  local label=$1 v=$2; shift 2
  local D=$PWD/dash-infer_amd/lib${v:+/$v}
  local r=$(env "$@" DIHIP_LIB_DIR=$D timeout 300 python tools/attn_block_trace.py 2>&1 | grep "one launch" | sed 's/.*: *//; s/ us per.*//')
  echo "$label lib=${v:-product} $* -> $r" | tee -a $OUT/sweep.txt
}
run base "" X=1
run base ab_nosweep X=1
run base ab_early8 X=1
for o in 10 20 25 30; do run napO "" DIHIP_AB_NAP_O=$o; done
for o in 20 30; do run napO ab_nosweep DIHIP_AB_NAP_O=$o; done
for q in 4 8 12; do run napQ "" DIHIP_AB_NAP_Q=$q; done
for m in 2 4 8; do run napM "" DIHIP_AB_NAP_M=$m; done
run spread "" DIHIP_AB_SPREAD=1
run combo "" DIHIP_AB_NAP_O=25 DIHIP_AB_NAP_Q=8 DIHIP_AB_SPREAD=1
run combo "" DIHIP_AB_NAP_O=25 DIHIP_AB_NAP_Q=8 DIHIP_AB_NAP_M=4 DIHIP_AB_SPREAD=1
run combo ab_nosweep DIHIP_AB_NAP_O=25 DIHIP_AB_NAP_Q=8
run base "" X=2
