#!/bin/bash
# One build->measure cycle on the GPU box: -m gpu tests, bench line, rocprofv3 kernel stats.  Everything logs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee gpurun_out/summary.log
tail -4 gpurun_out/pytest_gpu.log
grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head -20
if [ "$rc" != "0" ] && [ -z "$FORCE_BENCH" ]; then exit 0; fi
timeout -k 10 600 python bench.py --steps ${STEPS:-6} --warmup 2 ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench rc=$?" | tee -a gpurun_out/summary.log
tail -4 gpurun_out/bench.err
cat gpurun_out/bench.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 ${BENCH_ARGS} > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?" | tee -a $R/gpurun_out/summary.log
find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof/bench_kernel_stats.csv")))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 4 steps: %.1f" % (tot/1e6))
for r in rows[:22]:
    print("%-70s calls %5s  total %8.2f ms  avg %9.1f us  %5s%%" % (r["Name"].replace("(anonymous namespace)::","")[:70], r["Calls"], int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
