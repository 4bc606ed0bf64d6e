"""Solve times on the reference's small configurations (41^3 V2, 129^3 V3, 257^3 V5), beside the time per
cycle of bench.py's CPU baseline leg (the reference-order cycle on one host core) at the same size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multigrid_amd import libmg, problem
import bench

for n, depth in ((41, 2), (129, 3), (257, 5)):
    N = (n, n, n)
    dh = problem.main_extents_dh(N)
    rho = problem.synthetic_rho(N)
    with libmg.Handle(N, dh, depth) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        h.solve(3, 0.0)
        h.set_phi(np.zeros(N))
        h.reset_state()
        t0 = time.perf_counter()
        ok, cyc, l2 = h.solve(10000, 1e-6)
        dt = time.perf_counter() - t0
        dev = h.last_call_ms()
    cpu = bench.cpu_baseline(n, depth, 40 if n < 100 else 3)
    print(f"{n}^3 V{depth}: GPU {cyc} cycles {dt*1e3:.2f} ms wall ({dev:.2f} ms device, {dev/cyc*1e3:.0f} us/cycle) | "
          f"CPU lexicographic {cpu['s_per_cycle']*1e3:.2f} ms/cycle")
