#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -p no:cacheprovider -k "gemm or layernorm or colsum" > gpurun_out/pytest_gemm.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gemm.log
echo "=== new 256x128 3-stage"; timeout -k 10 300 python tools/gemm_bench.py 2>&1 | tee gpurun_out/gemm_new.log
echo "=== old 128x128 2-stage"; SC_GEMM_NT=128 timeout -k 10 300 python tools/gemm_bench.py 2>&1 | grep "^NT" | tee gpurun_out/gemm_old.log
