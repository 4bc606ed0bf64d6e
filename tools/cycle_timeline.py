#!/usr/bin/env python3
"""Kernel-by-kernel timeline of one V-cycle from a rocprofv3 --kernel-trace CSV directory
(the cycle before the last complete one)."""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the V-cycle's fused residual kernel (its last template argument, true, is the Gauss-Seidel solver's norm)
marks = [i for i, r in enumerate(rows) if re.search(r"k_residual_restrict<[^>]*false>\(", r["Kernel_Name"])]
a, b = marks[-3], marks[-2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].replace("mgk::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%9.1f dur=%7.1f grid=%7s wg=%4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"], k))
