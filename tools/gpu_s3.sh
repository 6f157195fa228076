#!/bin/bash
OUT=gpurun_out/${TAG:-r02e}
mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/dash-infer_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
for sh in ${SHAPES:-gate_up_swiglu down_addto}; do
  SHAPE=$sh TRACE=1 TRACE_BINS=1 timeout 120 ./tools/gemv_bench 4 128 1 2>&1 | grep -v "warm\|pre-trace\|trace launch" | tee -a $OUT/gemv_trace.txt
done
