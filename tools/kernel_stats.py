#!/usr/bin/env python
"""Print the top rows of a rocprofv3 *_kernel_stats.csv (found recursively when the exact path is absent)."""
import csv, glob, os, sys
path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
if not os.path.exists(path):
    c = glob.glob(os.path.join(os.path.dirname(path), "**", os.path.basename(path)), recursive=True)
    path = c[0] if c else path
rows = list(csv.DictReader(open(path)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("file %s: total kernel time %.2f ms" % (os.path.basename(path), tot / 1e6))
for r in rows[:top]:
    print("%-72s calls %6s  total %9.2f ms  avg %9.1f us  %5s%%" % (r["Name"].replace("(anonymous namespace)::", "")[:72], r["Calls"],
          int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
