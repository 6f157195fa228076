#!/bin/bash
# round 5, call i: the rest of the GPU suite after the MoE scratch-tensor fix
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5i
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8 | tee gpurun_out/r5i/pytest_gpu.log
