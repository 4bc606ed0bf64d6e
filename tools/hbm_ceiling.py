#!/usr/bin/env python3
"""Practical HBM rates of this GPU for the traffic mixes of the fine-grid kernels, measured with
plain elementwise torch kernels on arrays of the 513^3 field size (1.08 GB each): a fill (write
only), a sum reduction (read only), a copy (1 read : 1 write), out = a + b (2 reads : 1 write,
the mix of the red-black sweep) and a fill of three arrays from one (1 read : 3 writes, the mix of
the field kernel).  Prints one JSON line."""
import json

import torch

n = 513 ** 3
dev = torch.device("cuda", 0)
a = torch.rand(n, dtype=torch.float64, device=dev)
b = torch.rand(n, dtype=torch.float64, device=dev)
c = torch.empty_like(a)
c3 = torch.empty((n, 3), dtype=torch.float64, device=dev)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


gb = n * 8 / 1e9
out = {"bytes_per_array_GB": gb}
for name, fn, arrays in [("fill", lambda: c.fill_(1.0), 1), ("sum", lambda: a.sum(), 1),
                         ("copy", lambda: c.copy_(a), 2), ("add_2r1w", lambda: torch.add(a, b, out=c), 3),
                         ("expand_1r3w", lambda: c3.copy_(a[:, None].expand(n, 3)), 4),
                         ("fill3", lambda: c3.fill_(1.0), 3)]:
    t = timed(fn)
    out[name] = {"us": t * 1e6, "GBps": arrays * gb / t}
print(json.dumps(out))
