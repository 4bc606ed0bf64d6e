"""Kernel rates on the shape one rank of an 8-GPU 1025^3 run owns (128 x 1025 x 1025 planes),
measured on one GPU as a whole array of that shape: same point count as 513^3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multigrid_amd import libmg

shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 1025, 1025)
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dh = (1e-3, 1e-3, 1e-3)
g = np.random.default_rng(1)
h = libmg.Handle(shape, dh, depth)
h.set_rho(g.uniform(-1, 1, size=shape) * 8.85e-9)
h.set_phi(np.zeros(shape))
h.vcycles(2)
h.set_profiling(True)
h.reset_timers()
h.vcycles(5)
npts = float(np.prod(shape))
for k, t in h.timers().items():
    if t["launches"]:
        us = 1e3 * t["ms"] / t["launches"]
        print(f"{k:16s} n={t['launches']:4d} avg_us={us:9.1f}  ns/Mpt={1e3 * us / (npts / 1e6):8.2f}")
