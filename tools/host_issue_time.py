#!/usr/bin/env python3
"""Host time the slab driver needs to ENQUEUE one V-cycle (world size 1, RCCL backend) against the
time the GPU needs to run it: the driver stays asynchronous only while the first is smaller."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from multigrid_amd import dist as mgd, problem  # noqa: E402

for n in (513, 257):
    N = (n, n, n)
    depth = problem.reference_depth_for(N)
    s = mgd.DistSolver(N, problem.main_extents_dh(N), depth, mgd.HipOps(0), mgd.Comm())
    s.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
    s.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
    for _ in range(3):
        s.smooth_and_residual()
        s.coarse_correction()
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        s.smooth_and_residual()
        s.coarse_correction()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n}^3 V{depth}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms per cycle, GPU {1e3 * (t2 - t0) / K:.3f} ms per cycle",
          flush=True)
    del s
dist.destroy_process_group()
