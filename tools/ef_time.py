#!/usr/bin/env python3
"""Times mg_compute_ef on the device-resident 513^3 potential (HIP events inside the library).
usage: ef_time.py [n] [path of another build of libmg_hip.so to time instead]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import libmg, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
if len(sys.argv) > 2:
    libmg.LIB_PATH = os.path.abspath(sys.argv[2])
N = (n, n, n)
h = libmg.Handle(N, problem.main_extents_dh(N), 0)
h.set_phi(np.random.default_rng(1).random(N))
h.compute_ef(download=False)
ms = []
for _ in range(5):
    h.compute_ef(download=False)
    ms.append(h.last_call_ms())
print(f"{n}^3 compute_ef: best {min(ms):.3f} ms  {32.0 * n**3 / min(ms) / 1e6:.0f} GB/s   all {['%.3f' % m for m in ms]}")
