#!/bin/bash
# round 4, call ai: is it the `nt` bit or the asm issue pattern?  lib/ksl_nt = the product's asm loads of the activation fragments
# WITH the nt bit; product = without; stand-alone timing of the 7B gate/up pair at M = 32 and 16, alternating
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for rep in 1 2; do for v in ksl_nt product; do
  L=dash-infer_amd/lib; [ $v = ksl_nt ] && L=dash-infer_amd/lib/ksl_nt
  for m in 32 16; do
    printf "%-8s M=%d  " $v $m
    SHAPE=gate_up_swiglu LD_LIBRARY_PATH=$L timeout 120 ./tools/gemv_bench 4 128 $m 5 2>&1 | grep -v "warm-up" | tail -1 | sed 's/.*avg/avg/'
  done
done; done 2>&1 | tee gpurun_out/r4ai_kslice_nt_vs_asm.txt
