#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
timeout -k 10 ${BENCH_TIMEOUT:-600} python bench.py --steps ${STEPS:-3} --warmup ${WARMUP:-1} ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?" | tee gpurun_out/summary.log
cat gpurun_out/bench.err | tail -20
cat gpurun_out/bench.json
