#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -p no:cacheprovider -k "attention" > gpurun_out/pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_attn.log
echo "== workgroup per head (default)"; timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep attention | tee gpurun_out/attn_block.log
echo "== wave per head (SC_ATTENTION=wave)"; SC_ATTENTION=wave timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep attention | tee gpurun_out/attn_wave.log
