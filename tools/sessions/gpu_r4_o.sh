#!/bin/bash
# round 4, call o: what-if timing of the K-slice GEMM at M = 32 (gate/up pair incl. its RMSNorm launch): which part of the
# kernel the time goes to.  kxN = dash-infer_amd/lib/kxN built with -DDIHIP_KSL_X=N (1 no exchange/barrier, 2 no compute, 4 no x loads)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for v in ${VARIANTS:-"" kx1 kx2 kx3 kx4 kx7}; do
  [ "$v" = base ] && v=""
  for lr in 0 1; do
    printf "%-6s LR=%d  " "${v:-base}" $lr
    DIHIP_KSLICE_LDSRING=$lr SHAPE=gate_up_swiglu LD_LIBRARY_PATH=dash-infer_amd/lib/$v timeout 120 ./tools/gemv_bench 4 128 32 5 2>&1 | grep -v "warm-up" | tail -1 | sed 's/.*avg/avg/'
  done
done 2>&1 | tee -a gpurun_out/r4o_kslice_whatif.txt
