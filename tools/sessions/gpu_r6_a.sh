#!/bin/bash
# r6 a: the new bench line on hardware + the 1-rank RCCL tests
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_rccl_one_rank.py tests/test_bench_self_launch.py -q -x --timeout 300 2>&1 | tail -15 > $OUT/pytest.log
cat $OUT/pytest.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.out 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt
tail -c 4500 $OUT/bench_default.out
wc -c $OUT/bench_default.out
tail -5 $OUT/bench_default.err
tail -3 $OUT/bench_time.txt
python bench.py --gpus 2; echo "rc=$?"
