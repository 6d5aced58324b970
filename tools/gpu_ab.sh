#!/bin/bash
# A/B: GELU / GELU' as GEMM epilogues (default) vs separate HBM-bound kernels (SC_BLOCK_UNFUSE_GELU=1); then smoke().
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
: > gpurun_out/ab3.log
SC_BLOCK_UNFUSE_GELU=1 timeout -k 10 300 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 250 -p no:cacheprovider -k "bf16" 2>&1 | tail -2 | tee -a gpurun_out/ab3.log
for cfg in "SC_BLOCK_UNFUSE_GELU=0" "SC_BLOCK_UNFUSE_GELU=1" "SC_BLOCK_UNFUSE_GELU=0" "SC_BLOCK_UNFUSE_GELU=1"; do
  echo "== $cfg" | tee -a gpurun_out/ab3.log
  env $cfg timeout -k 10 200 python bench.py --steps 12 --warmup 3 --cpu-baseline 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab3.log
done
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/ab3.log
