#!/bin/bash
# Full GPU pass: all -m gpu tests, smoke, bench line, rocprofv3 kernel stats of the same command.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee gpurun_out/summary.log
tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/summary.log
tail -3 gpurun_out/smoke.log
timeout -k 10 900 python bench.py --steps ${STEPS:-4} --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench rc=$?" | tee -a gpurun_out/summary.log
tail -3 gpurun_out/bench.log
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?" | tee -a $R/gpurun_out/summary.log
ls -R $R/gpurun_out/prof | head -20
