"""Solve times on the reference's small configurations (41^3 V2, 129^3 V3, 257^3 V5) next to the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multigrid_amd import libmg, problem
from oracle import mg_oracle as orc

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
    phi = np.zeros(N)
    s = orc.Solver(N, dh, depth)
    t0 = time.perf_counter()
    ok2, cyc2, l22, _ = s.solve(phi, rho, 10000 if n < 257 else 5, 1e-6)
    dtc = time.perf_counter() - t0
    print(f"{n}^3 V{depth}: GPU {cyc} cycles {dt*1e3:.2f} ms wall ({dev:.2f} ms device, {dev/cyc*1e3:.0f} us/cycle) | "
          f"CPU lexicographic {cyc2} cycles {dtc:.3f} s ({dtc/cyc2*1e3:.2f} ms/cycle)")
