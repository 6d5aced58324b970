#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_quick.log | cut -c1-900
