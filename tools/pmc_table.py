#!/usr/bin/env python
"""Average rocprofv3 --pmc counter values per (kernel, grid) over the given output directories.  argv[1] = kernel-name filter."""
import collections, csv, glob, sys
flt = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if flt in r["Kernel_Name"]:
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
                agg[(name, r.get("Grid_Size", r.get("Grid_Size_X", "?")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), cs in sorted(agg.items()):
    vals = {k: sum(v) / len(v) for k, v in cs.items()}
    n = max(len(v) for v in cs.values())
    line = f"{name:48s} grid {grid:>8s} x{n:<3d} " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(vals.items()))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_BUSY_CYCLES" in vals and vals["SQ_BUSY_CYCLES"] > 0:
        # SQ_BUSY_CYCLES is summed over the SEs/XCDs it is sampled per; MFMA busy is per SIMD-cycle: report the raw ratio only
        line += f"  | MFMA_BUSY/BUSY_CYCLES={vals['SQ_VALU_MFMA_BUSY_CYCLES'] / vals['SQ_BUSY_CYCLES']:.3f}"
    if "FETCH_SIZE" in vals or "WRITE_SIZE" in vals:
        line += f"  | HBM-side bytes: read {2 * vals.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f} MB (FETCH_SIZE x2, gfx950) write {vals.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB"
    print(line)
