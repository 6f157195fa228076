#!/bin/bash
# One GPU-box session: parity tests (each file in its own process so that a fault in one does not
# hide the others), smoke, bench, and a rocprofv3 kernel trace of the bench.  Logs -> gpurun_out/.
# usage: tools/gpu_run.sh [tag]
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import torch; print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))" > $OUT/env.log 2>&1
nproc >> $OUT/env.log; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> $OUT/env.log
for f in test_gpu_gemm test_gpu_kv_attn test_gpu_glue test_gpu_prefill test_gpu_host_ops test_gpu_moe test_gpu_decoder test_gpu_tp_loopback; do
  timeout 900 python -m pytest tests/$f.py -q -m gpu -x --timeout 600 > $OUT/$f.log 2>&1
  echo "$f exit $?" | tee -a $OUT/summary.log
  tail -5 $OUT/$f.log
done
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.log; tail -3 $OUT/smoke.log
timeout 600 python bench.py --steps 64 --warmup 8 > $OUT/bench_int4_b1.json 2> $OUT/bench_int4_b1.err; echo "bench exit $?" | tee -a $OUT/summary.log
tail -c 3000 $OUT/bench_int4_b1.json; tail -5 $OUT/bench_int4_b1.err
if [ "${PROFILE:-1}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err)
  echo "rocprof exit $?" | tee -a $OUT/summary.log
  find $OUT/prof -name "*stats*" | head; find $OUT/prof -name "*kernel_stats*" -exec head -30 {} \;
  # keep only the small summaries
  find $OUT/prof -name "*kernel_trace*" -size +8M -delete
fi
for w in ${EXTRA_WORKLOADS:-}; do
  timeout 900 python bench.py --steps 32 --warmup 4 --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "bench $w exit $?" | tee -a $OUT/summary.log
  tail -c 2500 $OUT/bench_$w.json
done
timeout 200 python tools/prefill_bench.py 512 1024 2048 4096 8192 16384 > $OUT/prefill.txt 2>&1; tail -6 $OUT/prefill.txt
timeout 200 python tools/attn_batch_bench.py > $OUT/attn_batch.txt 2>&1; tail -6 $OUT/attn_batch.txt
cat $OUT/summary.log
