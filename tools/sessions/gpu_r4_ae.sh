#!/bin/bash
# round 4, call ae: heads per workgroup chunk of the matrix-core attention at batch 1 (DIHIP_ATTN_MFMA_HC; 16 = one chunk per KV group)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for hc in 16 8 4 2; do
  echo "== DIHIP_ATTN_MFMA_HC=$hc"
  DIHIP_ATTN_MFMA_HC=$hc python tools/attn_b1_bench.py 2>&1 | grep "us/layer"
done 2>&1 | tee gpurun_out/r4ae_attn_hc.txt
