"""Problem set-up shared by the benchmark, the smoke test and the host examples:
the reference driver's box and the deterministic synthetic charge density."""
import math

import numpy as np

EPS_0 = 8.85418782e-12  # reference cpp/World.h:14


def main_extents_dh(n):
    """Mesh spacing of the reference driver's domain: extents (-0.2,-0.2,-0.1)..(0.2,0.2,0.3)
    (reference cpp/Main.cpp:37), dh = (xm-x0)/(n-1) (cpp/World.cpp:29-30)."""
    x0 = (-0.2, -0.2, -0.1)
    xm = (0.2, 0.2, 0.3)
    return tuple((xm[a] - x0[a]) / (n[a] - 1) for a in range(3))


def synthetic_rho(n, i_range=None):
    """rho = EPS_0*1000*(sin(pi x)sin(pi y)sin(pi z) + 0.5 sin(5 pi x)sin(3 pi y)sin(7 pi z)),
    x,y,z = i/(ni-1), j/(nj-1), k/(nk-1); rho/EPS_0 ~ 1e3 V/m^2, comparable to the
    reference PIC run's q*n/EPS_0.  i_range=(i0,i1) returns only that slab of planes."""
    i0, i1 = (0, n[0]) if i_range is None else i_range
    x = np.arange(i0, i1) / (n[0] - 1)
    y = np.arange(n[1]) / (n[1] - 1)
    z = np.arange(n[2]) / (n[2] - 1)
    pi = math.pi
    a = np.sin(pi * x)[:, None, None] * np.sin(pi * y)[None, :, None] * np.sin(pi * z)[None, None, :]
    b = np.sin(5 * pi * x)[:, None, None] * np.sin(3 * pi * y)[None, :, None] * np.sin(7 * pi * z)[None, None, :]
    return np.ascontiguousarray(EPS_0 * 1000.0 * (a + 0.5 * b))


def reference_depth_for(n):
    """Deepest V-cycle whose coarsest grid still has >= 16 nodes per axis
    (513 -> 5 like MultiGridSolverV5, 129 -> 3, 41 -> 1)."""
    d = 1
    while (min(n) >> (d + 1)) >= 16:
        d += 1
    return d
