#!/usr/bin/env python3
"""Condenses a tools/profile.sh run (gpurun_out/<tag>/) into profiles/<tag>_*:
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
   <tag>_summary.json       per kernel: calls, avg/max duration, per-launch FETCH_SIZE / WRITE_SIZE
   <tag>_summary.md         the same as a table

HBM traffic follows the MI355X guide's recipe: FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE counts a 128-byte read request as
64 bytes, so for kernels whose loads are 16 bytes per lane (wide coalesced) the read side is
doubled.  Kernels with 8-byte-per-lane loads mix 64- and 128-byte requests and are
uncalibrated: both the raw and the doubled figure are listed for them.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# FETCH_SIZE tallies a 128-byte read request as 64 bytes on gfx950, so it is doubled before it is
# compared with a byte count.  Every kernel of this library reads whole lines (coalesced rows): a
# pass with TCC_EA0_RDREQ_32B/_64B/_128B (tools/profile_rd.sh, summarised by this script when its
# output is present) shows > 99.99 % of the read requests to be 128-byte ones in all of them.
FETCH_FACTOR = 2.0


def short(name):
    n = name.replace("mgk::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats_csv = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
    shutil.copy(stats_csv, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    kern = {}
    for r in csv.DictReader(open(stats_csv)):
        kern[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                  "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])}
    for which in ("fetch", "write"):
        f = glob.glob(os.path.join(src, which, "*", "*_counter_collection.csv"))
        if not f:
            continue
        vals = defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)  # KiB -> bytes
        for k, v in vals.items():
            if k in kern:
                kern[k][f"{which}_bytes_max"] = max(v)
                kern[k][f"{which}_bytes_avg"] = sum(v) / len(v)
    for k, d in kern.items():
        if "fetch_bytes_max" in d:
            d["fetch_correction"] = FETCH_FACTOR
            d["hbm_bytes_max_launch"] = FETCH_FACTOR * d["fetch_bytes_max"] + d.get("write_bytes_max", 0.0)
    # optional: exact read bytes from the request-size counters
    rd = glob.glob(os.path.join(src, "rd", "*", "*_counter_collection.csv"))
    if rd:
        per = defaultdict(lambda: defaultdict(dict))
        for r in csv.DictReader(open(rd[0])):
            per[short(r["Kernel_Name"])][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        for k, disp in per.items():
            if k not in kern:
                continue
            best = max(disp.values(), key=lambda v: v.get("TCC_EA0_RDREQ_sum", 0.0))
            n32, n64, n128 = (best.get(f"TCC_EA0_RDREQ_{w}B_sum", 0.0) for w in (32, 64, 128))
            kern[k]["read_requests_max_launch"] = {"32B": n32, "64B": n64, "128B": n128,
                                                   "bytes": 32 * n32 + 64 * n64 + 128 * n128}
    # the same key per grid size where a kernel runs on several shapes (slab runs: edge and interior launches)
    out = {"tag": tag, "kernels": kern}
    if os.path.exists(os.path.join(src, "command.txt")):   # tools/profile_slab.sh
        out["command"] = open(os.path.join(src, "command.txt")).read().strip()
        title = "one rank of a slab-partitioned cycle, alone on one MI355X"
    else:
        bench = json.load(open(os.path.join(src, "bench_stats.json")))
        out["command"] = ("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 "
                          "--cpu-cycles 0 --no-solve (513^3 V5); --pmc FETCH_SIZE / --pmc WRITE_SIZE in "
                          "separate runs with --steps 3")
        out["bench_under_profiler"] = {k: bench[k] for k in ("value", "ms_per_step", "roofline")}
        title = "513^3 V5 cycle, one MI355X"
    json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}` — {title}\n\n{out['command']}\n\n")
        f.write("| kernel | calls | avg us | max us | % time | FETCH_SIZE max (MB) | WRITE_SIZE max (MB) | "
                "HBM bytes/launch = 2 x FETCH + WRITE (MB) | read bytes from request sizes (MB) |\n|---|---|---|---|---|---|---|---|---|\n")
        for k, d in sorted(kern.items(), key=lambda kv: -kv[1]["pct"]):
            fm = d.get("fetch_bytes_max")
            wm = d.get("write_bytes_max")
            hb = d.get("hbm_bytes_max_launch")
            f.write(f"| {k} | {d['calls']} | {d['avg_us']:.1f} | {d['max_us']:.1f} | {d['pct']:.1f} | "
                    f"{fm / 1e6 if fm else 0:.1f} | {wm / 1e6 if wm else 0:.1f} | "
                    f"{hb / 1e6 if hb else 0:.1f} | "
                    f"{d['read_requests_max_launch']['bytes'] / 1e6 if 'read_requests_max_launch' in d else float('nan'):.1f} |\n")
    print(open(os.path.join(dst, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main(sys.argv[1])
