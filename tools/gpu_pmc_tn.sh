#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/pmc; export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -o p -- python3 $R/tools/gemm_one_tn.py > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run tn_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run tn_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
run tn_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
(export SC_GEMM_TN=128; run tn_old_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT)
python3 - <<PY
import csv, glob, collections
for d in ["tn_sq1","tn_sq2","tn_tcc","tn_old_sq2"]:
    files = glob.glob("$R/gpurun_out/pmc/%s/**/*counter_collection.csv" % d, recursive=True)
    if not files: print(d, "no counter file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if "gemm_bf16_tn" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, {k: sum(v)/len(v) for k, v in agg.items()})
PY
find $R/gpurun_out/pmc -name "*.csv" -size +5M -delete
