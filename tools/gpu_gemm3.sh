#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
export SC_GEMM_NT=3
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -p no:cacheprovider -k "gemm_bf16_nt" > gpurun_out/pytest_gemm.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gemm.log
timeout -k 10 300 python tools/gemm_bench.py 2>&1 | grep "^NT" | tee gpurun_out/gemm_v3.log
timeout -k 10 300 python tools/epi_bench.py 2>&1 | tee -a gpurun_out/gemm_v3.log
