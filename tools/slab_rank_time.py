#!/usr/bin/env python3
"""GPU time of ONE rank's share of the slab-partitioned V-cycle, measured on one GPU without peers.

The rank runs the schedule it would run in a world of W ranks (same slab, same edge / interior
launches, the whole coarse levels) with a communicator that moves nothing: halo planes keep
their initial zeros, so the numbers mean nothing, the kernels and their launch geometry are the
real ones.  What is missing from these times is only what the transport adds on a multi-GPU node.

The default driver is the product's one, the C++ rank inside libmg_hip (mg_dist with
MG_TRANSPORT_NONE), timed as a whole; --python runs the schedule's Python twin
(multigrid_amd/dist.py) instead, which also times the levels one by one.

    python tools/slab_rank_time.py [--size 1025] [--cases 8:3,8:0,4:1,2:0] [--cut 4] [--python]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import dist as mgd, problem  # noqa: E402


class SoloComm(mgd.Comm):
    """Rank `rank` of `world`, alone: every transfer is skipped, the all-gather keeps the rank's own part."""

    def __init__(self, rank, world):
        super().__init__()
        self.rank, self.world = rank, world

    def exchange(self, slab, own, width):
        slab.ready()

    def exchange_start(self, slab, own, width):
        slab.ready()

    def allreduce_sum(self, t):
        return t

    def allgather_planes(self, plan, l, kind, slab, full):
        lo, hi = plan.own(l, kind, self.rank)
        full.rows(lo, hi).copy_(slab.rows(lo, hi))


def measure(n, world, rank, cycles=10, overlap=True, cut=None):
    N = (n, n, n)
    depth = problem.reference_depth_for(N)
    s = mgd.DistSolver(N, problem.main_extents_dh(N), depth, mgd.HipOps(0), SoloComm(rank, world), overlap=overlap, cut=cut)
    s.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
    s.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
    marks = []

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    inner_sweeps = s._sweeps

    def timed_sweeps(l, *a, **k):
        e0 = ev()
        inner_sweeps(l, *a, **k)
        marks.append((f"sweeps_l{l}", e0, ev()))
    s._sweeps = timed_sweeps
    inner_tail = s._coarse_levels

    def timed_tail():
        e0 = ev()
        inner_tail()
        marks.append(("tail", e0, ev()))
    s._coarse_levels = timed_tail

    for _ in range(2):
        s.smooth_and_residual()
        s.coarse_correction()
    torch.cuda.synchronize()
    marks.clear()
    phases = []
    for _ in range(cycles):
        a = ev()
        s.smooth_and_residual()
        b = ev()
        s.coarse_correction()
        c = ev()
        phases.append((a, b, c))
    torch.cuda.synchronize()
    out = {"world": world, "rank": rank, "planes": s.own(0, "phi"), "cut": s.cut,
           "cycle_ms": sum(a.elapsed_time(c) for a, b, c in phases) / cycles,
           "smooth_residual_ms": sum(a.elapsed_time(b) for a, b, c in phases) / cycles,
           "coarse_ms": sum(b.elapsed_time(c) for a, b, c in phases) / cycles}
    for name in sorted({m[0] for m in marks}):
        out[name + "_ms"] = sum(e0.elapsed_time(e1) for nm, e0, e1 in marks if nm == name) / cycles
    del s
    torch.cuda.empty_cache()
    return out


def measure_cxx(n, world, rank, cycles=10, overlap=True, cut=None, flags=0, deep_halos=True):
    """The same share driven by mg_dist (the product's driver): device time of `cycles` cycles."""
    from multigrid_amd import libmg
    N = (n, n, n)
    depth = problem.reference_depth_for(N)
    d = libmg.DistRank(N, problem.main_extents_dh(N), depth, rank, world, device=0, coarse_cut=cut or 0,
                       overlap=overlap, alone=True, flags=flags, deep_halos=deep_halos)
    d.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
    d.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
    d.vcycles(2)
    d.vcycles(cycles)
    out = {"world": world, "rank": rank, "planes": d.own, "cut": d.cut, "deep_halos": d.deep_halos,
           "cycle_ms": d.stats()["last_ms"] / cycles}
    d.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1025)
    ap.add_argument("--cases", default="8:3,8:0,4:1,2:0")
    ap.add_argument("--no-overlap", action="store_true", help="whole-slab launches, no edge/interior split")
    ap.add_argument("--cut", type=int, default=None, help="first level solved whole on rank 0 (default: first with <= 128 planes)")
    ap.add_argument("--python", action="store_true", help="time the Python twin of the schedule, level by level")
    ap.add_argument("--fast", action="store_true", help="MG_FLAG_FAST_SMOOTHER (C++ driver only)")
    ap.add_argument("--no-deep", action="store_true", help="an exchange after every sweep on every level (C++ driver only)")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    for case in args.cases.split(","):
        w, r = (int(x) for x in case.split(":"))
        if args.python:
            res = measure(args.size, w, r, overlap=not args.no_overlap, cut=args.cut)
        else:
            from multigrid_amd import libmg
            res = measure_cxx(args.size, w, r, overlap=not args.no_overlap, cut=args.cut,
                              flags=libmg.MG_FLAG_FAST_SMOOTHER if args.fast else 0, deep_halos=not args.no_deep)
        print(" ".join(f"{k}={v:.3f}" if isinstance(v, float) else f"{k}={v}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
