#!/bin/bash
TAG=${1:-attn}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kv_attn.py tests/test_gpu_glue.py -q -m gpu -x --timeout 600 > $OUT/test_attn.log 2>&1; echo "test_attn exit $?"; tail -15 $OUT/test_attn.log
