#!/bin/bash
# A/B of scheduling knobs on the full training step (bench.py, 12 timed steps each, no CPU baseline / roofline replay differences).
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
: > gpurun_out/ab.log
for cfg in "SC_BLOCK_DW_GATE=1 SC_GEMM_NT_SPLIT=1" "SC_BLOCK_DW_GATE=0 SC_GEMM_NT_SPLIT=1" "SC_BLOCK_DW_GATE=1 SC_GEMM_NT_SPLIT=0" "SC_BLOCK_DW_GATE=0 SC_GEMM_NT_SPLIT=0" "SC_BLOCK_DW_GATE=1 SC_GEMM_NT_SPLIT=1"; do
  echo "== $cfg" | tee -a gpurun_out/ab.log
  env $cfg timeout -k 10 200 python bench.py --steps 12 --warmup 3 --cpu-baseline 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab.log
done
timeout -k 10 400 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -3 | tee -a gpurun_out/ab.log
