#!/bin/bash
# Bisect a non-bit-identical full-size step by kernel selection.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
: > gpurun_out/bisect.log
for cfg in "SC_BLOCK_FUSE_CS=0" "SC_BLOCK_FUSE_CS=0" "SC_BLOCK_FUSE_CS=0" "A=1"; do
  echo "== [$cfg]" >> gpurun_out/bisect.log
  env $cfg timeout -k 10 200 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 180 -p no:cacheprovider -k "full_size_step_properties" 2>&1 | grep -E "passed|failed|diff:" >> gpurun_out/bisect.log
done
cat gpurun_out/bisect.log
