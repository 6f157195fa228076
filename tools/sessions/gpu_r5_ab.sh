#!/bin/bash
# round 5, call ab: tensor-parallel decode through the C++ operator layer, a rank per thread on one GPU (P2P all-reduce between the threads' streams)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5ab
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
for a in "2 none 1 4 128" "4 none 2 4 128" "8 none 1 4 128" "2 i8 3 8 -1"; do
  echo "== hostdecode $a"
  timeout 240 python tests/p2p_worker.py hostdecode $a 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-600
done 2>&1 | tee gpurun_out/r5ab/log.txt
