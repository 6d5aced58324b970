#!/bin/bash
# Kernel-time distribution of the ViT-L/14 (BASELINE config 5) step, local batch 512.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/prof_c5; export TMPDIR=/tmp; cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -o c5 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --experiment experiment_10 --model ViT-L-14 --local-batch 512 > $R/gpurun_out/c5prof.log 2>&1
echo "rc=$?"
find $R/gpurun_out/prof_c5 -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_c5/c5_kernel_stats.csv")))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 3 steps: %.1f" % (tot/1e6))
for r in rows[:16]:
    print("%-64s calls %5s  total %8.2f ms  avg %9.1f us  %5s%%" % (r["Name"].replace("(anonymous namespace)::","")[:64], r["Calls"], int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
