#!/usr/bin/env python3
"""Wave-cycle breakdown per kernel and grid size from a tools/profile_sq.sh run (gpurun_out/<tag>/sq)."""
import collections
import csv
import glob
import sys

f = glob.glob(f"gpurun_out/{sys.argv[1]}/sq/*/*_counter_collection.csv")[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("mgk::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    d[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | grid | waves | parked | issue-stalled | issuing | waiting on LDS | VALU instr/wave | LDS instr/wave |")
print("|---|---|---|---|---|---|---|---|---|")
for (k, g), c in sorted(d.items(), key=lambda kv: -sum(kv[1]["SQ_WAVE_CYCLES"])):
    a = {n: sum(v) / len(v) for n, v in c.items()}
    wc = a["SQ_WAVE_CYCLES"]
    if wc < 1e6:
        continue
    print("| %s | %s | %d | %.2f | %.2f | %.2f | %.2f | %d | %d |" % (
        k, g, a["SQ_WAVES"], a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc,
        a["SQ_WAIT_INST_LDS"] / wc, a["SQ_INSTS_VALU"] / a["SQ_WAVES"], a["SQ_INSTS_LDS"] / a["SQ_WAVES"]))
