#!/bin/bash
# Two ranks on ONE GPU over gloo: rehearses bench.py's N > 1 path (shards, embedding all-gather, bucketed SUM all-reduce, MAX timing).
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R; export TMPDIR=/tmp
export SC_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --local-batch 256 --cpu-baseline 0 > gpurun_out/dp2.json 2> gpurun_out/dp2.err
echo "dp2 rc=$?"; tail -5 gpurun_out/dp2.err | cut -c1-300; cat gpurun_out/dp2.json | cut -c1-600
unset SC_DIST_BACKEND
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 2 --warmup 1 --local-batch 256 --cpu-baseline 0 > gpurun_out/dp1.json 2> gpurun_out/dp1.err
echo "dp1 rc=$?"; cat gpurun_out/dp1.json | cut -c1-400
