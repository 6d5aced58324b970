#!/bin/bash
# PMC passes (counters only + kernel-trace, one group per run) on the NT GEMM; summaries printed and kept under gpurun_out/pmc.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/pmc; export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -o p -- python3 $R/tools/gemm_one.py > $R/gpurun_out/pmc/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections
for d in ["sq1","sq2","tcc1","fetch","write"]:
    files = glob.glob("$R/gpurun_out/pmc/%s/**/*counter_collection.csv" % d, recursive=True)
    if not files: print(d, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        if "gemm_bf16" in r["Kernel_Name"]:
            agg[(r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X","?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for grid, cs in agg.items():
        print(d, "grid", grid, {k: sum(v)/len(v) for k, v in cs.items()})
PY
find $R/gpurun_out/pmc -name "*.csv" -size +5M -delete
