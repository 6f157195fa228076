#!/bin/bash
# A/B build of a few translation units: tools/build_ksl_variant.sh NAME "-DFLAG ..." [PREFIX] compiles the PREFIX*.hip units
# (default gemm_kslice_: the K-slice GEMM instantiations) with the extra flags into dash-infer_amd/lib/NAME/ and links them
# with the product build's other objects; load it with
# DIHIP_LIB_DIR=$PWD/dash-infer_amd/lib/NAME (python) or LD_LIBRARY_PATH (tools/gemv_bench).  Timing experiments only.
set -e
cd "$(dirname "$0")/../dash-infer_amd/csrc"
NAME=$1; FLAGS=$2; PREFIX=${3:-gemm_kslice_}; OUT=../lib/$NAME
mkdir -p $OUT/obj
for f in ${PREFIX}*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -fno-gpu-rdc -ffp-contract=off $FLAGS -c $f -o $OUT/obj/${f%.hip}.o &
done
wait
OBJS=$(ls ../lib/obj/*.o | grep -v /${PREFIX})
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdashinfer_hip.so $OBJS $OUT/obj/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
ls -la $OUT/libdashinfer_hip.so
