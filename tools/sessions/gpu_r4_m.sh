#!/bin/bash
# round 4, call m: BASELINE configs[4] (Qwen2-57B-A14B MoE, int8) decode step through BOTH runners -- decoder.DecodeSession and
# the operator layer (fusion pass -> DihipMoeBlock, hipGraph replay) -- one bench line
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 800 python bench.py --workload cfg5_moe --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4m_bench_cfg5_moe.json 2> gpurun_out/r4m_bench_cfg5_moe.err
echo "rc $?"; tail -c 3000 gpurun_out/r4m_bench_cfg5_moe.json; tail -5 gpurun_out/r4m_bench_cfg5_moe.err
