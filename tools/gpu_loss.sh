#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 300 python tools/loss_bench.py 2>&1 | tee gpurun_out/loss_bench.log
