#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command (program itself after --, no wrappers).
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
timeout -k 10 ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps ${STEPS:-3} --warmup 1 --cpu-baseline 0 ${BENCH_ARGS} > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?" | tee $R/gpurun_out/summary.log
tail -3 $R/gpurun_out/rocprof.log
find $R/gpurun_out/prof -name "*stats*" | head; 
F=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); head -40 "$F"
# keep the merged output small: drop the per-dispatch trace, keep the stats
find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
