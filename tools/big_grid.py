#!/usr/bin/env python3
"""One GPU holding a whole large grid: the V-cycle and a solve to 1e-6 at n^3 (default 2049^3 with
seven coarse grids: 3 x 68.8 GB of fine fields + 3 x 9.9 GB of coarse ones = 236 GB of the GPU's
288 GB).  The host arrays are filled a few planes at a time; prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import libmg, problem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = (n, n, n)
depth = problem.reference_depth_for(N)
t0 = time.perf_counter()
h = libmg.Handle(N, problem.main_extents_dh(N), depth)
print(f"handle for {n}^3 V{depth} created in {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
t0 = time.perf_counter()
rho = np.empty(N)
for lo in range(0, n, 16):
    hi = min(lo + 16, n)
    rho[lo:hi] = problem.synthetic_rho(N, (lo, hi))
    if lo % 256 == 0:
        print(f"rho planes {lo}.. filled", file=sys.stderr, flush=True)
print(f"rho filled in {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
t0 = time.perf_counter()
h.set_rho(rho)
t_rho = time.perf_counter() - t0
del rho
phi0 = np.zeros(N)
t0 = time.perf_counter()
h.set_phi(phi0)
t_phi = time.perf_counter() - t0
del phi0
print(f"uploads: rho {t_rho:.1f} s, phi {t_phi:.1f} s", file=sys.stderr, flush=True)
h.vcycles(2)
l2 = h.vcycles(cycles)
ms = h.last_call_ms() / cycles
print(f"{ms:.2f} ms per cycle", file=sys.stderr, flush=True)
out = {"grid": n, "depth": depth, "ms_per_cycle": ms, "gdof_per_s": float(n) ** 3 / ms / 1e6,
       "upload_rho_s": t_rho, "upload_phi_s": t_phi, "L2_after_%d_cycles" % (cycles + 2): l2}
if "--solve" in sys.argv:
    t0 = time.perf_counter()
    ok, cyc, l2s = h.solve(400, 1e-6)
    out["solve"] = {"converged": ok, "more_cycles": cyc, "L2": l2s, "seconds": time.perf_counter() - t0}
h.close()
print(json.dumps(out))
