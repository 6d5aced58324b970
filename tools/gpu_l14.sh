#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 400 -p no:cacheprovider -k "l14 or towers_forward" > gpurun_out/pytest_l14.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/pytest_l14.log | cut -c1-400
