#!/bin/bash
# The two-workgroup-per-CU AGPR kernel (SC_GEMM_NT=p) against the default dispatch: correctness tests, then per-shape throughput.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
SC_GEMM_NT=p timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -p no:cacheprovider -k "gemm_bf16_nt" > gpurun_out/pytest_pp.log 2>&1; echo "pytest(pp) rc=$?"; tail -3 gpurun_out/pytest_pp.log
echo "== SC_GEMM_NT=p"; SC_GEMM_NT=p timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep "^NT" | tee gpurun_out/gemm_pp.log
echo "== default"; timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep "^NT" | tee gpurun_out/gemm_default.log
