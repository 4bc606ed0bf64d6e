#!/usr/bin/env python3
"""Exact against fast smoother (MG_FLAG_FAST_SMOOTHER), kernel by kernel: per-category HIP-event
timers of the V-cycle, the plain fine sweep and the sweep pair (through mg_gs_solve)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import libmg, problem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
N = (n, n, n)
depth = problem.reference_depth_for(N)
dh = problem.main_extents_dh(N)
rho = problem.synthetic_rho(N)
for name, flags in (("exact", 0), ("fast", libmg.MG_FLAG_FAST_SMOOTHER)):
    h = libmg.Handle(N, dh, depth, flags=flags)
    h.set_rho(rho)
    h.set_phi(np.zeros(N))
    h.vcycles(3)
    h.vcycles(10)
    ms = h.last_call_ms() / 10
    h.set_profiling(True)
    h.reset_timers()
    h.vcycles(5)
    t = h.timers()
    line = " ".join(f"{k}={1e3 * v['ms'] / 5:.0f}us/{v['launches'] // 5}" for k, v in t.items() if v["launches"])
    h.reset_timers()
    for _ in range(4):
        h.gs_solve(1, 0.0)
    one = 1e3 * h.timers()["smooth_fine"]["ms"] / 4
    h.reset_timers()
    h.gs_solve(25, 0.0)
    pair = (1e3 * h.timers()["smooth_fine"]["ms"] - one) / 12
    print(f"{name:6s} cycle {ms:.3f} ms | per cycle: {line} | plain sweep {one:.0f} us, sweep pair {pair:.0f} us", flush=True)
    h.close()
