#!/usr/bin/env python3
"""Benchmark of the multigrid V-cycle on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one full V-cycle iteration (smooth, residual + norm, restrict, coarse
levels, prolong + correct) on the synthetic charge density of
multigrid_amd/problem.py, with phi and rho resident in HBM before the timed
region.  N = 1 runs the 513^3 five-level cycle (the configuration the headline
metric is quoted on).  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries:
  roofline      the dominant kernel's algorithmic bytes / its HIP-event time, vs 8 TB/s
  cpu_baseline  the reference algorithm (lexicographic GS-SOR cycle; oracle/) timed
                single-threaded on this host on a bounded number of cycles
  solve         cycles and seconds to residual norm < 1e-6 from phi = 0
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

# Algorithmic (compulsory) bytes per point of the level a kernel runs on, fp64.
KERNEL_BYTES_PER_POINT = {"smooth_fine": 16.0, "residual": 24.0, "restrict": 9.0,
                          "smooth_coarse": 16.0, "prolong": 9.0, "prolong_correct": 17.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=0, help="fine grid nodes per axis (default 513)")
    ap.add_argument("--depth", type=int, default=0, help="coarse grids (default: 5 at 513)")
    ap.add_argument("--cpu-cycles", type=int, default=3, help="cycles of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-solve", action="store_true", help="skip the time-to-1e-6 solve")
    ap.add_argument("--tol", type=float, default=1e-6)
    return ap.parse_args()


def cpu_baseline(n, depth, cycles):
    """The reference algorithm on one host core, same inputs, `cycles` iterations."""
    import numpy as np
    from oracle import mg_oracle as orc
    orc.build()
    N = (n, n, n)
    dh = orc.main_extents_dh(N)
    rho = orc.synthetic_rho(N)
    phi = np.zeros(N)
    s = orc.Solver(N, dh, depth)
    t0 = time.perf_counter()
    s.solve(phi, rho, cycles, 0.0)
    dt = time.perf_counter() - t0
    per = dt / cycles
    return {"value": 1.0 / per, "unit": "V-cycles/s", "cores": 1, "kind": "port",
            "sample": f"{cycles} iterations of the lexicographic (reference-order) {n}^3 V{depth} cycle, "
                      f"gcc -O3 single thread, host has {os.cpu_count()} logical CPUs",
            "s_per_cycle": per, "gdof_per_s": n ** 3 / per / 1e9}


def run_single(args):
    import numpy as np
    from multigrid_amd import libmg, problem

    libmg.load()
    if libmg.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible (the HIP path has no CPU fallback)")
    n = args.size or 513
    N = (n, n, n)
    depth = args.depth or problem.reference_depth_for(N)
    dh = problem.main_extents_dh(N)
    rho = problem.synthetic_rho(N)
    npts = float(n) ** 3

    h = libmg.Handle(N, dh, depth, device=0)
    h.set_rho(rho)
    h.set_phi(np.zeros(N))
    if args.warmup:
        h.vcycles(args.warmup)
    t0 = time.perf_counter()
    l2 = h.vcycles(args.steps)          # K iterations, no host sync inside; returns after a stream sync
    wall = time.perf_counter() - t0
    dev_ms = h.last_call_ms()
    ms_per_step = wall * 1e3 / args.steps
    vps = args.steps / wall

    # per-kernel HIP-event timing of the same cycle (separate pass: the events perturb the stream)
    h.set_profiling(True)
    h.reset_timers()
    h.vcycles(max(3, min(args.steps, 10)))
    timers = h.timers()
    h.set_profiling(False)
    kernels = {}
    for k, t in timers.items():
        if t["launches"]:
            kernels[k] = {"launches": t["launches"], "avg_us": 1e3 * t["ms"] / t["launches"],
                          "GBps": t["bytes"] / (t["ms"] * 1e-3) / 1e9 if t["ms"] > 0 else None,
                          "share": None}
    tot = sum(t["ms"] for t in timers.values()) or 1.0
    for k in kernels:
        kernels[k]["share"] = timers[k]["ms"] / tot
    dom = max(kernels, key=lambda k: kernels[k]["share"])
    roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS, "traffic": None,
                "avg_us": kernels[dom]["avg_us"],
                "bytes_per_launch": timers[dom]["bytes"] / timers[dom]["launches"]}

    solve = None
    if not args.no_solve:
        h.set_phi(np.zeros(N))
        h.reset_state()
        t0 = time.perf_counter()
        ok, cycles, l2s = h.solve(10000, args.tol)
        dt = time.perf_counter() - t0
        solve = {"tol": args.tol, "converged": ok, "cycles": cycles, "seconds": dt, "L2": l2s}
    h.close()

    cpu = cpu_baseline(n, depth, args.cpu_cycles) if args.cpu_cycles > 0 else None

    out = {"metric": "V-cycles/s", "value": vps, "unit": "V-cycles/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{n}^3 fp64 multigrid V{depth} cycle (sweeps {list(h.cfg.nu[:depth + 1])}, omega 1.4), "
                                  f"red-black GS-SOR, 1 GPU", "grid": n, "depth": depth},
           "gdof_per_s": npts * vps / 1e9, "device_ms_per_step": dev_ms / args.steps, "last_L2": l2,
           "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu, "solve": solve}
    if cpu:
        out["speedup_vs_cpu_per_cycle"] = cpu["s_per_cycle"] / (wall / args.steps)
    print(json.dumps(out))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus == 1 and world == 1:
        run_single(args)
        return
    raise SystemExit("bench.py: the multi-GPU (slab-partitioned) path is not implemented yet")


if __name__ == "__main__":
    main()
