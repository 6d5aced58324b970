#!/bin/bash
# First GPU pass: kernel-level parity, model-level parity, smoke, short bench.  Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 240 -p no:cacheprovider > gpurun_out/pytest_kernels.log 2>&1
echo "kernels rc=$?" | tee -a gpurun_out/summary.log
tail -5 gpurun_out/pytest_kernels.log
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 240 -p no:cacheprovider -s > gpurun_out/pytest_model.log 2>&1
echo "model rc=$?" | tee -a gpurun_out/summary.log
tail -5 gpurun_out/pytest_model.log
