#!/bin/bash
# round 5, call ae: a model whose weights come from a serialized weight file (.asparam written by the reference's own writer)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r5ae
timeout 200 python -m pytest tests/test_gpu_host_ops.py -q -m gpu --timeout 150 -k "serialized_file" 2>&1 | tail -15 | cut -c1-400 | tee gpurun_out/r5ae/log.txt
