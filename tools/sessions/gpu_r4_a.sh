#!/bin/bash
# round 4, first GPU call: the new host runner tests, the full-depth parity tests, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/a_device.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_host_runner.py -x -q -s > gpurun_out/a_host_runner.log 2>&1; echo "host_runner rc=$?" >> gpurun_out/a_rc.txt
timeout 1500 python -m pytest tests/test_gpu_parity_depth.py -q -s > gpurun_out/a_parity_depth.log 2>&1; echo "parity_depth rc=$?" >> gpurun_out/a_rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/a_bench_default.json 2> gpurun_out/a_bench_default.err; echo "bench rc=$?" >> gpurun_out/a_rc.txt
tail -5 gpurun_out/a_host_runner.log; tail -12 gpurun_out/a_parity_depth.log; cat gpurun_out/a_rc.txt; head -c 1500 gpurun_out/a_bench_default.json
