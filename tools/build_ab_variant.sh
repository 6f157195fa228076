#!/bin/bash
# A/B build of the fused attention block: only decode_attn_block.hip is recompiled with the extra flags, the other objects are the
# product build's.   tools/build_ab_variant.sh <name> [-DDIHIP_AB_EARLY=8 ...] [TRACE=1 in the environment: stamps compiled in]
#   -> dash-infer_amd/lib/<name>/libdashinfer_hip.so, loaded with DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/<name>
set -e
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/dash-infer_amd/csrc
BASE=$ROOT/dash-infer_amd/lib${TRACE:+/trace}
OUT=$ROOT/dash-infer_amd/lib/$NAME
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$SRC -Wno-unused-result -fno-gpu-rdc -ffp-contract=off \
  ${TRACE:+-DDIHIP_GEMV_TRACE=1} "$@" -c $SRC/decode_attn_block.hip -o $OUT/decode_attn_block.o
OBJS=$(ls $BASE/obj/*.o | grep -v decode_attn_block.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdashinfer_hip.so $OBJS $OUT/decode_attn_block.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built $OUT/libdashinfer_hip.so
