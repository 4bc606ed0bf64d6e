"""mg_dev_residual_restrict_noalloc -- the entry point that allocates nothing -- against the
two-call form it is a variant of (mg_dev_residual_restrict with a caller-owned partial-sum buffer,
then mg_dev_sum_partials): same restricted residual and same sum of squares, bit for bit, on two
calls in a row, the second on a smaller grid (the built-in workspace keeps the first call's
contents).  The fixed reference for both is the CPU oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from util import rng, phi_like, rho_like, assert_bit_equal, EPS_0

pytestmark = pytest.mark.gpu


def _run(gpu, orc, shape, g):
    lib = gpu.load()
    dev = torch.device("cuda", 0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    n = tuple(shape)
    dh = orc.main_extents_dh(n)
    coef = orc.coeffs(dh, 0)
    phi = phi_like(n, g)
    b = rho_like(n, g) / EPS_0
    pitch = (n[2] + 1) & ~1
    nc = tuple(a >> 1 for a in n)
    cpitch = (nc[2] + 1) & ~1

    def dev_padded(a, p):
        t = torch.zeros(a.shape[:2] + (p,), dtype=torch.float64, device=dev)
        t[:, :, :a.shape[2]] = torch.from_numpy(a)
        return t

    d_phi, d_b = dev_padded(phi, pitch), dev_padded(b, pitch)
    out = {}
    for name in ("two_call", "noalloc"):
        d_rc = torch.full((nc[0], nc[1], cpitch), 7.0, dtype=torch.float64, device=dev)
        total = torch.full((1,), -1.0, dtype=torch.float64, device=dev)
        arrs = [gpu.MgDevArray(t.data_ptr(), t.shape[0], t.shape[1], k, 0)
                for t, k in ((d_phi, n[2]), (d_b, n[2]), (d_rc, nc[2]))]
        args = [stream] + [C.byref(a) for a in arrs] + [(C.c_int64 * 3)(*n), (C.c_double * 4)(*coef)]
        if name == "two_call":
            partials = torch.zeros(1 << 16, dtype=torch.float64, device=dev)
            count = C.c_int64(0)
            gpu._check(lib.mg_dev_residual_restrict(*args, C.c_void_p(partials.data_ptr()), partials.numel(), 0, n[0],
                                                    C.byref(count)))
            gpu._check(lib.mg_dev_sum_partials(stream, C.c_void_p(partials.data_ptr()), count.value,
                                               C.c_void_p(total.data_ptr())))
        else:
            gpu._check(lib.mg_dev_residual_restrict_noalloc(*args, 0, n[0], C.c_void_p(total.data_ptr())))
        torch.cuda.synchronize()
        out[name] = (d_rc.cpu().numpy()[:, :, :nc[2]].copy(), float(total.item()))
    assert_bit_equal(out["noalloc"][0], out["two_call"][0], f"restricted residual at {n}")
    assert np.float64(out["noalloc"][1]).view(np.uint64) == np.float64(out["two_call"][1]).view(np.uint64)
    # and both against the oracle
    R, s = orc.residual(phi, b * EPS_0, coef)
    want = orc.restrict27(R, np.full(nc, 7.0))
    assert_bit_equal(out["noalloc"][0], want, f"restricted residual at {n} against the oracle")
    assert out["noalloc"][1] == pytest.approx(s, rel=1e-12)


def test_noalloc_equals_two_call_form_twice_in_a_row(gpu, orc):
    g = rng(77)
    _run(gpu, orc, (129, 129, 129), g)
    _run(gpu, orc, (41, 37, 45), g)   # smaller, ragged: fewer partial sums than the workspace still holds


def test_noalloc_call_after_a_warm_up_leaves_free_memory_unchanged(gpu):
    """After one warm-up call on the device, a call must not change the device's free memory (the
    first call may: it can make the runtime load the library's code object, workspace included)."""
    lib = gpu.load()
    dev = torch.device("cuda", 0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    n, nc = (65, 65, 65), (32, 32, 32)
    d_phi = torch.rand((65, 65, 66), dtype=torch.float64, device=dev)
    d_b = torch.rand((65, 65, 66), dtype=torch.float64, device=dev)
    d_rc = torch.zeros((32, 32, 32), dtype=torch.float64, device=dev)
    total = torch.zeros(1, dtype=torch.float64, device=dev)
    arrs = [gpu.MgDevArray(t.data_ptr(), t.shape[0], t.shape[1], k, 0) for t, k in ((d_phi, 65), (d_b, 65), (d_rc, 32))]
    args = [stream] + [C.byref(a) for a in arrs] + [(C.c_int64 * 3)(*n), (C.c_double * 4)(1.0, 1.0, 1.0, 6.0), 0, 65,
                                                    C.c_void_p(total.data_ptr())]
    gpu._check(lib.mg_dev_residual_restrict_noalloc(*args))  # warm-up
    torch.cuda.synchronize()
    first = float(total.item())
    free0, _ = torch.cuda.mem_get_info(dev)
    for _ in range(3):
        gpu._check(lib.mg_dev_residual_restrict_noalloc(*args))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free1 == free0
    assert float(total.item()) == first and first > 0


def test_noalloc_rejects_what_does_not_fit(gpu):
    lib = gpu.load()
    arr = gpu.MgDevArray(256, 4, 65535, 65535, 0)  # never dereferenced: the size check comes first
    total = torch.zeros(1, dtype=torch.float64, device="cuda")
    rc = lib.mg_dev_residual_restrict_noalloc(None, C.byref(arr), C.byref(arr), C.byref(arr),
                                              (C.c_int64 * 3)(65535, 65535, 65535), (C.c_double * 4)(1, 1, 1, 6),
                                              0, 4, C.c_void_p(total.data_ptr()))
    assert rc == -1  # MG_ERR_INVALID
    assert b"workspace" in lib.mg_last_error()
