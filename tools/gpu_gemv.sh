#!/bin/bash
# GEMV iteration loop on the GPU box: parity tests of the GEMM family + stand-alone shape timings.
TAG=${1:-gemv}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export LD_LIBRARY_PATH=$PWD/dash-infer_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu -x --timeout 600 > $OUT/test_gpu_gemm.log 2>&1; echo "test_gpu_gemm exit $?"; tail -5 $OUT/test_gpu_gemm.log
fi
for cfg in "4 128 1" "8 -1 1" ${EXTRA_CFG:-}; do
  timeout 300 ./tools/gemv_bench $cfg 2>&1 | tee -a $OUT/gemv_bench.log
done
if [ "${OLD:-0}" = "1" ]; then
  echo "--- general kernel (DIHIP_GEMV_STREAM=0)" | tee -a $OUT/gemv_bench.log
  DIHIP_GEMV_STREAM=0 timeout 300 ./tools/gemv_bench 4 128 1 2>&1 | tee -a $OUT/gemv_bench.log
fi
