#!/bin/bash
# Round-end evidence: PMC traffic of the GEMM launch set, then the standard cycle (all -m gpu tests, bench line, rocprofv3 kernel stats).
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/gpu_pmc_traffic.sh
cp $R/gpurun_out/gemm_traffic.json $R/profiles/_traffic_new.json 2>/dev/null
cd $R && bash tools/gpu_cycle.sh
