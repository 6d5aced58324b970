#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
SC_ATTENTION_SHORT=2 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -p no:cacheprovider -k "attention" > gpurun_out/pytest_attn.log 2>&1; echo "pytest(short2) rc=$?"; tail -3 gpurun_out/pytest_attn.log
echo "== default"; timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep attention | tee gpurun_out/attn_default.log
echo "== SC_ATTENTION_SHORT=2 (recompute kernels)"; SC_ATTENTION_SHORT=2 timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep attention | tee gpurun_out/attn_short2.log
