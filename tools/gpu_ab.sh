#!/bin/bash
# A/B of scheduling knobs on the full training step (bench.py, 12 timed steps each).
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
: > gpurun_out/ab2.log
for cfg in "SC_STREAM_PRIO=0" "SC_STREAM_PRIO=t" "SC_STREAM_PRIO=s" "SC_STREAM_PRIO=0" "SC_STREAM_PRIO=t" "SC_STREAM_PRIO=s"; do
  echo "== $cfg" | tee -a gpurun_out/ab2.log
  env $cfg timeout -k 10 200 python bench.py --steps 12 --warmup 3 --cpu-baseline 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab2.log
done
