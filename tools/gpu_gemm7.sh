#!/bin/bash
# Tile-walk cell of the NT 256x256 kernel (row tiles x column tiles per XCD cell): per-shape throughput for a few shapes.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
: > gpurun_out/gemm_group.log
for g in "8 4" "4 8" "16 2" "4 4" "2 16" "32 1" "8 2"; do
  set -- $g
  echo "== GROUP_M=$1 GROUP_N=$2" | tee -a gpurun_out/gemm_group.log
  SC_GEMM_NT_GROUP_M=$1 SC_GEMM_NT_GROUP_N=$2 timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep "^NT" | grep "img.qkv\|img.fc1 +bias\|img.fc2 \[\|img.dx_fc1\|txt.fc1 +bias\|txt.fc2 \[" | tee -a gpurun_out/gemm_group.log
done
