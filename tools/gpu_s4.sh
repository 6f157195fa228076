#!/bin/bash
OUT=gpurun_out/${TAG:-r02g}
mkdir -p $OUT
for i in 1 2 3; do timeout 300 python -m pytest "tests/test_gpu_gemm.py::test_batch_invariance_and_determinism_full_size" -q -m gpu -x 2>&1 | tail -2; done
export LD_LIBRARY_PATH=$PWD/dash-infer_amd/lib/trace:/opt/rocm/lib:$LD_LIBRARY_PATH
for sh in ${SHAPES:-qkv_norm_gemv o_addto gate_up_swiglu down_addto}; do
  SHAPE=$sh TRACE=1 TRACE_BINS=1 timeout 120 ./tools/gemv_bench 4 128 1 2>&1 | grep -v "warm\|pre-trace\|trace launch\|\.\.\s*[0-9]" | tee -a $OUT/gemv_trace.txt
done
