"""Parity of the rows next to the V-cycle: the electric field E = -grad(phi)
(reference PotentialSolver::computeEF, cpp/PotentialSolver.cpp:418-458) and the plain
Gauss-Seidel/SOR solver (GaussSeidelSolver::solve, cpp/PotentialSolver.cpp:235-291), both
through the C ABI against the CPU oracle.  Bit equality for the fields; the norm is a tree sum
on the GPU and a sequential one in the oracle (relative tolerance 1e-11, as for the V-cycle)."""
import numpy as np
import pytest

from util import assert_bit_equal, phi_like, rel_l2, rng

pytestmark = pytest.mark.gpu

EF_SHAPES = [(3, 3, 3), (4, 5, 6), (17, 17, 17), (20, 20, 20), (33, 21, 45), (41, 41, 41),
             (64, 64, 64), (65, 65, 65), (5, 200, 129), (129, 129, 129)]


@pytest.mark.parametrize("shape", EF_SHAPES, ids=str)
def test_compute_ef_operator(gpu, orc, shape):
    phi = phi_like(shape, rng(21))
    dh = (0.4 / (shape[0] - 1), 0.37 / (shape[1] - 1), 0.41 / (shape[2] - 1))
    assert_bit_equal(gpu.op_compute_ef(phi, dh), orc.compute_ef(phi, dh), "ef")


@pytest.mark.parametrize("scale", [0.0, 1e-320, 1e-300, 1e-250, 1e250, 1e305], ids=str)
def test_compute_ef_extreme_magnitudes(gpu, orc, scale):
    """Zeros keep their sign (-(0 - 0)/(2h) = -0), subnormal and near-overflow quotients round like
    the division instruction: the range where the reciprocal-FMA form is not used."""
    shape = (9, 20, 70)
    phi = phi_like(shape, rng(22)) * 1e-3 * scale
    phi[2:5, 3:9, 10:40] = 0.0
    dh = (0.01, 0.02, 0.03)
    with np.errstate(over="ignore", invalid="ignore"):
        want = orc.compute_ef(phi, dh)
    assert_bit_equal(gpu.op_compute_ef(phi, dh), want, "ef")


def test_smoother_extreme_magnitudes(gpu, orc):
    """The same for the division in the smoother update."""
    shape = (12, 40, 70)
    n = shape
    c = orc.coeffs((0.01, 0.02, 0.03), 0)
    for scale in (0.0, 1e-318, 1e-290, 1e290):
        g = rng(23)
        u, rhs = phi_like(shape, g) * 1e-3 * scale, phi_like(shape, g) * 1e-3 * scale * c[3]
        u[3:6, 5:20, 8:30] = 0.0
        rhs[3:6, 5:20, 8:30] = 0.0
        with np.errstate(over="ignore", invalid="ignore"):
            want = orc.smooth(u.copy(), rhs, n, c, 1.4, 2, rb=True)
        got = gpu.op_smooth_rb(u.copy(), rhs, n, c, 1.4, 2)
        assert_bit_equal(got, want, f"smoother at scale {scale}")


@pytest.mark.parametrize("flags", [0, 4], ids=["ef_array_on_first_use", "ef_array_reserved"])
def test_compute_ef_on_the_handle_follows_the_solve(gpu, orc, flags):
    """mg_compute_ef works on the device-resident phi a solve left behind, with its output array
    allocated by the first call or, with MG_FLAG_RESERVE_EF, by mg_create."""
    assert gpu.MG_FLAG_RESERVE_EF == 4
    N = (41, 41, 41)
    dh, rho = orc.main_extents_dh(N), orc.synthetic_rho(N)
    with gpu.Handle(N, dh, 2, flags=flags) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        with pytest.raises(gpu.MgError):   # nothing computed yet
            h.get_ef()
        h.vcycles(3)                       # leaves through the pending-correction path
        ef = h.compute_ef()
        phi = h.get_phi()
        assert_bit_equal(ef, orc.compute_ef(phi, dh), "ef after cycles")
        h.solve(100, 1e-6)
        with pytest.raises(gpu.MgError):   # stale after a solve
            h.get_ef()
        assert_bit_equal(h.compute_ef(), orc.compute_ef(h.get_phi(), dh), "ef after solve")


@pytest.mark.parametrize("n,max_it", [((17, 17, 17), 60), ((41, 41, 41), 130), ((33, 21, 45), 77),
                                      ((129, 129, 129), 30), ((140, 150, 160), 27),
                                      # batches of sweeps between two norm tests: 1 | 1 + 1 | 1 + 25 | 1 + 25 + 25
                                      ((129, 129, 129), 1), ((129, 129, 129), 2), ((130, 131, 135), 26),
                                      ((129, 140, 131), 51)], ids=str)
def test_gs_sweeps_match_rb_oracle_bit_for_bit(gpu, orc, n, max_it):
    dh, rho = orc.main_extents_dh(n), orc.synthetic_rho(n)
    want = np.zeros(n)
    w_ok, w_it, w_l2 = orc.gs_solve(want, rho, dh, max_it, 0.0, rb=True)
    with gpu.Handle(n, dh, 0) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(n))
        ok, its, l2 = h.gs_solve(max_it, 0.0)
        got = h.get_phi()
    assert (ok, its) == (w_ok, w_it) == (False, max_it)
    assert_bit_equal(got, want, "GS phi")
    assert l2 == pytest.approx(w_l2, rel=1e-11)


def test_gs_converges_like_the_reference_loop(gpu, orc):
    """Same stopping rule (test every 25th sweep, interior-only norm); the red-black iterate
    converges to the same discrete solution as the reference's lexicographic loop."""
    n = (33, 33, 33)
    dh, rho = orc.main_extents_dh(n), orc.synthetic_rho(n)
    lex = np.zeros(n)
    l_ok, l_it, _ = orc.gs_solve(lex, rho, dh, 20000, 1e-6, rb=False)
    twin = np.zeros(n)
    t_ok, t_it, t_l2 = orc.gs_solve(twin, rho, dh, 20000, 1e-6, rb=True)
    with gpu.Handle(n, dh, 0) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(n))
        ok, its, l2 = h.gs_solve(20000, 1e-6)
        got = h.get_phi()
        with pytest.raises(gpu.MgError):   # depth 0: no V-cycle on this handle
            h.solve(1, 1e-6)
    assert l_ok and t_ok and ok
    assert its == t_it and its % 25 == 1     # converged in a tested iteration
    assert_bit_equal(got, twin, "GS converged phi")
    assert l2 == pytest.approx(t_l2, rel=1e-11)
    assert rel_l2(got, lex) < 1e-6


def test_gs_on_a_multigrid_handle(gpu, orc):
    """mg_gs_solve shares phi/rho with the V-cycle handle (the reference's MG classes derive
    from GaussSeidelSolver)."""
    n = (41, 41, 41)
    dh, rho = orc.main_extents_dh(n), orc.synthetic_rho(n)
    want = np.zeros(n)
    orc.gs_solve(want, rho, dh, 10, 0.0, rb=True)
    with gpu.Handle(n, dh, 2) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(n))
        h.gs_solve(10, 0.0)
        assert_bit_equal(h.get_phi(), want, "GS on MG handle")


@pytest.mark.parametrize("lo,hi", [(0, 7), (6, 14), (13, 21), (0, 21)], ids=str)
def test_compute_ef_on_a_slab(gpu, orc, lo, hi):
    """mg_dev_compute_ef on planes [lo,hi) of a slab that stores one halo plane per side (plus
    what the one-sided face formulas reach) equals the rows of the whole-array result."""
    import ctypes as C
    import torch
    n = (21, 30, 37)
    dh = (0.01, 0.02, 0.03)
    phi = phi_like(n, rng(24))
    want = orc.compute_ef(phi, dh)[lo:hi]
    s_lo, s_hi = max(lo - 1, 0), min(hi + 1, n[0])
    if lo == 0:
        s_hi = max(s_hi, 3)
    if hi == n[0]:
        s_lo = min(s_lo, n[0] - 3)
    pitch = (n[2] + 1) & ~1
    dev = torch.device("cuda", 0)
    slab = torch.zeros((s_hi - s_lo, n[1], pitch), dtype=torch.float64, device=dev)
    slab[:, :, :n[2]] = torch.from_numpy(phi[s_lo:s_hi]).to(dev)
    ef = torch.full((hi - lo, n[1], n[2], 3), float("nan"), dtype=torch.float64, device=dev)
    arr = gpu.MgDevArray(slab.data_ptr(), slab.shape[0], n[1], n[2], s_lo)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    gpu._check(gpu.load().mg_dev_compute_ef(stream, C.byref(arr), C.c_void_p(ef.data_ptr()), lo,
                                           (C.c_int64 * 3)(*n), (C.c_double * 3)(*dh), lo, hi))
    torch.cuda.synchronize(dev)
    assert_bit_equal(ef.cpu().numpy(), want, "slab ef")
    # a slab without the halo plane is refused, not read out of bounds
    if lo > 0:
        short = gpu.MgDevArray(slab.data_ptr(), slab.shape[0], n[1], n[2], s_lo + 1)
        with pytest.raises(gpu.MgError):
            gpu._check(gpu.load().mg_dev_compute_ef(stream, C.byref(short), C.c_void_p(ef.data_ptr()), lo,
                                                   (C.c_int64 * 3)(*n), (C.c_double * 3)(*dh), lo, hi))


B_SHAPES = [(4, 5, 6), (17, 17, 17), (20, 20, 20), (33, 21, 45), (64, 64, 64), (129, 129, 129)]


@pytest.mark.parametrize("shape", B_SHAPES, ids=str)
def test_restrict7_operator(gpu, orc, shape):
    """The "B" classes' 7-point restriction (cpp/PotentialSolver.cpp:704-715)."""
    g = rng(31)
    Rf = phi_like(shape, g)
    Rc0 = phi_like(tuple(s >> 1 for s in shape), g)  # entries the loop does not reach must survive
    assert_bit_equal(gpu.op_restrict7(Rf, Rc0.copy()), orc.restrict7(Rf, Rc0.copy()), "restrict7")


@pytest.mark.parametrize("n,depth", [(41, 1), (41, 2), (129, 3), (129, 4), (257, 5)])
def test_b_variant_cycles_match_rb_oracle_bit_for_bit(gpu, orc, n, depth):
    N = (n, n, n)
    dh, rho = orc.main_extents_dh(N), orc.synthetic_rho(N)
    cycles = 3
    want = np.zeros(N)
    s = orc.Solver(N, dh, depth, variant_b=True)
    ok, cyc, l2, hist = s.solve(want, rho, cycles, 0.0, rb=True)
    with gpu.Handle(N, dh, depth, variant_b=True) as h:
        assert list(h.cfg.nu[:depth + 1]) == list(orc.schedule(depth, variant_b=True))
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        got_hist = [h.vcycles(1) for _ in range(cycles)]
        assert_bit_equal(h.get_phi(), want, "phi (B variant)")
        for l in range(1, depth + 1):
            assert_bit_equal(h.level_array("R", l), s.R(l), f"R_{l} (B variant)")
    assert got_hist == pytest.approx(list(hist), rel=1e-11)


def test_b_variant_converges_to_the_same_solution(gpu, orc):
    N = (65, 65, 65)
    dh, rho = orc.main_extents_dh(N), orc.synthetic_rho(N)
    ref = np.zeros(N)
    assert orc.Solver(N, dh, 3, variant_b=True).solve(ref, rho, 10000, 1e-6)[0]   # lexicographic, B
    with gpu.Handle(N, dh, 3, variant_b=True) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        ok, cyc, l2 = h.solve(10000, 1e-6)
        assert ok
        assert rel_l2(h.get_phi(), ref) <= 1e-6
