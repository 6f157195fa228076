#!/bin/bash
# round 4, call n: K-slice GEMM with the LDS-DMA weight ring at M = 32 -- parity (bit-identical to the register ring), then the
# stand-alone timing of the MLP shapes with either ring
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -k "kslice" -q -x -m gpu 2>&1 | tail -8
for lr in 0 1; do
  echo "== DIHIP_KSLICE_LDSRING=$lr"
  for shp in gate_up_swiglu down_addto; do
    DIHIP_KSLICE_LDSRING=$lr SHAPE=$shp LD_LIBRARY_PATH=dash-infer_amd/lib timeout 120 ./tools/gemv_bench 4 128 32 5 2>&1 | grep -v "warm-up" | tail -2
  done
  DIHIP_KSLICE_LDSRING=$lr SHAPE=gate_up_swiglu LD_LIBRARY_PATH=dash-infer_amd/lib timeout 120 ./tools/gemv_bench 8 -1 32 5 2>&1 | grep -v "warm-up" | tail -1
done 2>&1 | tee gpurun_out/r4n_kslice_ldsring.txt
