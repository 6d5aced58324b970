#!/bin/bash
# One-GPU bench lines for the other BASELINE configurations (per-GPU shards), plus the ViT-L/14 step test.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 300 -p no:cacheprovider -k "c5_vit" > gpurun_out/pytest_c5.log 2>&1; echo "c5 test rc=$?"; tail -3 gpurun_out/pytest_c5.log | cut -c1-600
run() { name=$1; shift; timeout -k 10 500 python bench.py --steps 4 --warmup 2 --cpu-baseline 0 "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; echo "$name rc=$?"; tail -2 gpurun_out/bench_$name.err | cut -c1-200; cut -c1-700 gpurun_out/bench_$name.json; echo; }
run c2 --experiment experiment_2 --local-batch 512
run c3 --experiment experiment_4 --local-batch 512
run c5 --experiment experiment_10 --model ViT-L-14 --local-batch 512
