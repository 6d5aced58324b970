#!/bin/bash
# A/B of the NT 256x256 kernel's de-phasing delay (SC_GEMM_STAGGER = multiplier of the quarter-tile unit; 0 = off)
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
for s in 0 1 2; do
  echo "== SC_GEMM_STAGGER=$s" | tee -a gpurun_out/gemm_stagger.log
  SC_GEMM_STAGGER=$s timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep "^NT" | grep "qkv\|fc1" | tee -a gpurun_out/gemm_stagger.log
done
