#!/bin/bash
# Kernel-by-kernel timeline (start, end, queue) of one V-cycle of one rank's share of the slab
# solve, from a rocprofv3 kernel trace of tools/slab_rank_time.py.  usage: [SIZE=2049] tools/slab_timeline.sh <tag> [case]
tag=${1:-slabtl}; case_=${2:-8:3}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/slab_rank_time.py --size ${SIZE:-1025} --cases $case_ > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_sum_partials" in r["Kernel_Name"]]
a, b = marks[-3] + 1, marks[-2] + 1
t0 = int(rows[a]["Start_Timestamp"])
with open(sys.argv[1] + "/timeline.txt", "w") as o:
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = r["Kernel_Name"].replace("mgk::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        o.write("%9.1f -> %9.1f dur=%7.1f q=%s grid=%7s wg=%4s %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Grid_Size_X"], r["Workgroup_Size_X"], k))
print(open(sys.argv[1] + "/timeline.txt").read())
PY
