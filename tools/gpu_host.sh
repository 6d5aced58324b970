#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_model.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_model.log
timeout -k 10 300 python tools/host_overhead.py 2>&1 | tail -1 | tee gpurun_out/host_overhead.log
