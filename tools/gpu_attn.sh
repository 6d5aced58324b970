#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -p no:cacheprovider -k "attention" > gpurun_out/pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_attn.log | cut -c1-700
