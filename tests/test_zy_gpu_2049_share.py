"""One rank's share of the 2049^3 grid split over 8 GPUs (256 fine planes of 2049 x 2049 nodes plus
halo planes, 3 x 8.7 GB on the device): the two fine-grid launches of the cycle -- the single-pass
red-black sweep and the fused residual + restriction -- run on a middle rank's slab through the
mg_dev_* entry points on random data and must equal the CPU oracle's slab-form operators bit for
bit.  The whole 2049^3 grid needs eight GPUs; this is the part of it one GPU of the node holds.
Collected late: it needs about 45 GB of host memory and a minute of oracle time."""
import ctypes as C

import numpy as np
import pytest
import torch

from util import EPS_0

pytestmark = pytest.mark.gpu

N0 = 2049
WORLD, RANK = 8, 3
DEPTH = 7


def _fill(shape, scale, seed):
    """Uniform(-scale, scale) without a second full-size temporary."""
    g = np.random.default_rng(seed)
    a = np.empty(shape)
    for i in range(0, shape[0], 16):
        blk = a[i:i + 16]
        g.random(out=blk)
        blk *= 2.0 * scale
        blk -= scale
    return a


def _equal_bits(got, want):
    return bool((got.view(np.uint64) == want.view(np.uint64)).all())


def test_middle_rank_share_of_2049_cubed(gpu, orc):
    from multigrid_amd import dist as mgd
    plan = mgd.SlabPlan((N0,) * 3, DEPTH, WORLD)
    assert plan.cut == 4 and plan.bounds[RANK + 1] - plan.bounds[RANK] == 256
    own = plan.own(0, "phi", RANK)
    ulo, uhi = plan.stored(0, "phi", RANK, 2)
    blo, bhi = plan.stored(0, "R", RANK, 1)
    clo, chi = plan.stored(1, "R", RANK, 1)
    n = (N0,) * 3
    pitch = (N0 + 1) & ~1
    free, _ = torch.cuda.mem_get_info()
    need = 8.0 * pitch * N0 * ((uhi - ulo) * 2 + (bhi - blo)) + 8.0 * 1026 * 1024 * (chi - clo)
    if free < 1.05 * need:
        pytest.skip(f"needs {need / 1e9:.1f} GB of free device memory")
    dh = orc.main_extents_dh(n)
    coef = orc.coeffs(dh, 0)
    lib = gpu.load()
    dev = torch.device("cuda", 0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    # host slabs in the device layout [plane][row][pitch]; the padding column stays zero
    u = np.zeros((uhi - ulo, N0, pitch))
    u[:, :, :N0] = _fill((uhi - ulo, N0, N0), 1e3, 1)
    b = np.zeros((bhi - blo, N0, pitch))
    b[:, :, :N0] = _fill((bhi - blo, N0, N0), 1e3 * 1e3, 2)  # rho/EPS_0-like magnitudes
    d_u = torch.from_numpy(u).to(dev)
    d_alt = torch.zeros_like(d_u)
    d_b = torch.from_numpy(b).to(dev)

    def arr(t, i0, nk=N0):
        return gpu.MgDevArray(t.data_ptr(), t.shape[0], t.shape[1], nk, i0)

    # ---- one full red-black sweep of the owned planes, out of place
    flag = C.c_int(0)
    gpu._check(lib.mg_dev_smooth_rb(stream, C.byref(arr(d_u, ulo)), C.byref(arr(d_alt, ulo)), C.byref(arr(d_b, blo)),
                                    (C.c_int64 * 3)(*n), (C.c_double * 4)(*coef), 1.4, 1, own[0], own[1], 0, 0,
                                    C.byref(flag)))
    assert flag.value == 1, "the single-pass out-of-place sweep is expected at this size"
    torch.cuda.synchronize()
    got = d_alt[own[0] - ulo:own[1] - ulo].cpu().numpy()
    want = u.copy()
    orc.lib().orc_slab_smooth_rb(C.byref(orc.view(want, N0, ulo)), C.byref(orc.view(b, N0, blo)),
                                 (C.c_int64 * 3)(*n), (C.c_double * 4)(*coef), 1.4, own[0], own[1])
    assert _equal_bits(got[:, :, :N0], want[own[0] - ulo:own[1] - ulo, :, :N0]), "sweep of the slab differs from the oracle"
    del got, want, d_alt

    # ---- fused residual + restriction of the owned planes
    nc = N0 >> 1
    cpitch = (nc + 1) & ~1
    d_rc = torch.zeros((chi - clo, nc, cpitch), dtype=torch.float64, device=dev)
    partials = torch.zeros(1 << 16, dtype=torch.float64, device=dev)
    total = torch.zeros(1, dtype=torch.float64, device=dev)
    count = C.c_int64(0)
    gpu._check(lib.mg_dev_residual_restrict(stream, C.byref(arr(d_u, ulo)), C.byref(arr(d_b, blo)),
                                            C.byref(arr(d_rc, clo, nc)), (C.c_int64 * 3)(*n), (C.c_double * 4)(*coef),
                                            C.c_void_p(partials.data_ptr()), partials.numel(), own[0], own[1],
                                            C.byref(count)))
    gpu._check(lib.mg_dev_sum_partials(stream, C.c_void_p(partials.data_ptr()), count.value,
                                       C.c_void_p(total.data_ptr())))
    torch.cuda.synchronize()
    rc_want = np.zeros((chi - clo, nc, cpitch))
    s_want = orc.lib().orc_slab_residual_restrict(C.byref(orc.view(u, N0, ulo)), C.byref(orc.view(b, N0, blo)),
                                                  C.byref(orc.view(rc_want, nc, clo)), (C.c_int64 * 3)(*n),
                                                  (C.c_double * 4)(*coef), own[0], own[1])
    colo, cohi = own[0] >> 1, own[1] >> 1
    got_rc = d_rc.cpu().numpy()
    assert _equal_bits(got_rc[colo - clo:cohi - clo, :, :nc], rc_want[colo - clo:cohi - clo, :, :nc]), \
        "restricted residual of the slab differs from the oracle"
    # 1.07e9 squares: the oracle adds them one by one (rounding error up to ~sqrt(N) ulp), the
    # device in a tree of partial sums; the restricted residual above is the bit-exact check
    assert float(total.item()) == pytest.approx(s_want, rel=1e-10)
