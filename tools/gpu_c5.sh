#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 300 -p no:cacheprovider -k "c5_vit or l14" 2>&1 | tail -2
timeout -k 10 500 python bench.py --steps 4 --warmup 2 --cpu-baseline 0 --experiment experiment_10 --model ViT-L-14 --local-batch 512 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; echo "c5 rc=$?"; cut -c1-260 gpurun_out/bench_c5.json
