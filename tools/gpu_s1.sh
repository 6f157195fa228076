#!/bin/bash
# round-2 session 1: codec parity re-check + the measurements that decide the batch-1 design
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$PWD/dash-infer_amd/lib:/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_gpu_kv_attn.py -q -m gpu -x --timeout 300 > $OUT/test_kv.log 2>&1; echo "kv tests exit $?"; tail -3 $OUT/test_kv.log
timeout 200 ./tools/mall_bench > $OUT/mall_bench.txt 2>&1; echo "mall exit $?"; cat $OUT/mall_bench.txt
PREFETCH=1 TRACE=1 timeout 300 ./tools/gemv_bench 4 128 1 > $OUT/gemv_bench.txt 2>&1; echo "gemv exit $?"; grep -v "warm\|pre-trace\|trace launch" $OUT/gemv_bench.txt | head -80
for ns in 0 8 4; do
  echo "--- attention pair, DIHIP_ATTN_NSPLITS=$ns" | tee -a $OUT/attn_bench.txt
  DIHIP_ATTN_NSPLITS=$ns timeout 120 ./tools/attn_bench 1 2048 0 2>&1 | tail -12 | tee -a $OUT/attn_bench.txt
done
