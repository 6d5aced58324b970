#!/bin/bash
# HBM traffic of the NT GEMM launch set: two separate PMC passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only.
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/pmc; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc/t_$c -o p -- python3 $R/tools/gemm_replay.py > $R/gpurun_out/pmc/t_$c.log 2>&1; echo "$c rc=$?"
done
python3 - <<PY
import csv, glob, json
tot = {}
n = 0
for c in ["FETCH_SIZE", "WRITE_SIZE"]:
    f = glob.glob("$R/gpurun_out/pmc/t_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_bf16_nt" in r["Kernel_Name"] and r["Counter_Name"] == c]
    tot[c] = sum(vals); n = len(vals)
# rocprofv3 reports KiB; gfx950: FETCH_SIZE counts half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section).
# One logical launch (sc_gemm_bf16_nt call) may be two kernels (256x256 kernel + 256x128 kernel on the leftover rows): the
# traffic is summed over all NT kernels and divided by the number of CALLS, the unit bench.py's roofline uses.
import sys
sys.path.insert(0, "$R")
import bench
from sparsify_clip_amd.model import CONFIGS
cfg = CONFIGS["ViT-B-32"]
calls = sum(r for _, _, r in bench.step_gemm_launches(cfg, 1024, 3 * cfg["patch"] ** 2, cfg["image_size"] // cfg["patch"]))
kernels = n
n = calls
hbm = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / n
out = {"launches": n, "kernels": kernels, "local_batch": 1024, "fetch_kib_sum": tot["FETCH_SIZE"], "write_kib_sum": tot["WRITE_SIZE"], "hbm_bytes_per_launch": hbm,
       "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, averaged over the launches of tools/gemm_replay.py"}
print(json.dumps(out))
json.dump(out, open("$R/gpurun_out/gemm_traffic.json", "w"))
PY
find $R/gpurun_out/pmc -name "*.csv" -size +5M -delete
