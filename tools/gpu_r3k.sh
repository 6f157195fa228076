#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3k; mkdir -p $OUT; cd $ROOT
( time timeout 1500 python -m pytest tests -m gpu -q --durations=15 ) > $OUT/pytest_gpu.log 2>&1
tail -30 $OUT/pytest_gpu.log | cut -c1-300
