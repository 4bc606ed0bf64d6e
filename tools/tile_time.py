#!/usr/bin/env python3
"""Time of one red-black sweep launch (mg_dev_smooth_rb, one sweep per call) on whole arrays or thin
slabs of given shapes: tools/tile_time.py planes,nj,nk[,lo,hi] ...   (lo,hi: plane range swept)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import libmg  # noqa: E402

lib = libmg.load()
SW = int(os.environ.get("SWEEPS", "5"))  # sweeps per call: amortises the host's cost of a call
dev = torch.device("cuda", 0)
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]
    planes, nj, nk = v[:3]
    lo, hi = (v[3], v[4]) if len(v) == 5 else (0, planes)
    pitch = (nk + 1) & ~1
    u = torch.rand((planes, nj, pitch), dtype=torch.float64, device=dev)
    alt = u.clone()
    rhs = torch.rand((planes, nj, pitch), dtype=torch.float64, device=dev)
    n = (C.c_int64 * 3)(max(planes, hi + 2) if len(v) == 5 else planes, nj, nk)
    coef = (C.c_double * 4)(1.0, 1.0, 1.0, 6.0)
    arr = lambda t: libmg.MgDevArray(t.data_ptr(), planes, nj, nk, 0)
    flag = C.c_int(0)
    bufs = [u, alt]

    def launch():
        libmg._check(lib.mg_dev_smooth_rb(stream, C.byref(arr(bufs[0])), C.byref(arr(bufs[1])), C.byref(arr(rhs)), n, coef,
                                          1.4, SW, lo, hi, 0, 0, C.byref(flag)))
        if flag.value:
            bufs.reverse()
    for _ in range(5):
        launch()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 100
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        launch()
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{spec:24s} {ev[0].elapsed_time(ev[1]) * 1e3 / reps / SW:8.2f} us per sweep  (out of place: {flag.value})", flush=True)
