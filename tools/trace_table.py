#!/usr/bin/env python3
"""Per-kernel, per-grid-size timing table from a rocprofv3 --kernel-trace CSV directory."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("mgk::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    d[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 50 or "lds" in k:
        print("%-44s grid=%8d n=%4d avg=%8.1f us total=%9.1f us" % (k, g, len(v), sum(v) / len(v), sum(v)))
