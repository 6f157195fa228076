#!/bin/bash
# round 5, call s: stand-alone timing of the deferred-norm kernels
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5s
{ for m in 32 16; do timeout 300 python tools/defer_norm_bench.py $m 2>&1 | grep -v amdgpu.ids; done; } 2>&1 | tee gpurun_out/r5s/log.txt
