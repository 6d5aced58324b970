#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 600 -p no:cacheprovider -k "c5_vit_l14_real" --durations=6 > gpurun_out/pytest_real.log 2>&1; echo "rc=$?"; tail -14 gpurun_out/pytest_real.log | cut -c1-300
