#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
for n in 1536 512; do
  echo "== SC_GEMM_NT_BIG_MINN=$n" | tee -a gpurun_out/gemm_minn.log
  SC_GEMM_NT_BIG_MINN=$n timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep "^NT" | grep -v "qkv \[\|fc1" | tee -a gpurun_out/gemm_minn.log
done
