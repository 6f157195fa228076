#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/r3d; mkdir -p $OUT; cd $ROOT
( time timeout 600 python -m pytest tests/test_gpu_host_graph.py -m gpu -q -s ) > $OUT/pytest_host_graph.log 2>&1
tail -25 $OUT/pytest_host_graph.log | cut -c1-400
( time timeout 900 python -m pytest tests/test_gpu_gemv_stress.py tests/test_gpu_host_ops.py tests/test_gpu_tp_loopback.py tests/test_gpu_p2p_processes.py -m gpu -q ) > $OUT/pytest_stress_tp.log 2>&1
tail -25 $OUT/pytest_stress_tp.log | cut -c1-400
