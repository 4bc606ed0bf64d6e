"""MG_FLAG_FAST_SMOOTHER: the red-black smoother of the coarse levels (l >= 1) with its point update
as one chain of fused multiply-adds on coefficients that carry the division and omega (8
operations instead of 22); the fine-grid sweep keeps the reference's expression.

The reference has no such form (its update is cpp/PotentialSolver.cpp:3780-3786), so what pins it
is (1) the oracle's twin of the SAME formulation, against which every kernel that smooths must
agree bit for bit, whole cycle states included, and (2) the bar that matters for a smoother: the
converged potential equals the reference-order (lexicographic, exact-update) solve to <= 1e-6
relative L2 at the same residual tolerance -- the residual itself is always evaluated exactly."""
import os

import numpy as np
import pytest

from util import assert_bit_equal, rel_l2, rng

pytestmark = pytest.mark.gpu


def problem(orc, n):
    N = (n, n, n) if isinstance(n, int) else tuple(n)
    return N, orc.main_extents_dh(N), orc.synthetic_rho(N)


@pytest.mark.parametrize("n,depth", [(41, 1), (41, 2), (65, 3), (129, 3), (129, 5), (257, 5), ((70, 141, 133), 2)], ids=str)
def test_fast_cycles_match_the_oracles_fast_twin_bit_for_bit(gpu, orc, n, depth):
    """phi, R_l and eps_l after three iterations: every smoothing kernel (the marching sweeps and
    sweep pairs, the on-load forms, the tiles, the coarsest level's LDS kernel) runs the short form
    and must reproduce the scalar CPU evaluation of that form."""
    N, dh, rho = problem(orc, n)
    k = 3
    phi = np.zeros(N)
    s = orc.Solver(N, dh, depth)
    _, _, _, hist = s.solve(phi, rho, k, 0.0, fast=True)
    exact = np.zeros(N)
    orc.Solver(N, dh, depth).solve(exact, rho, k, 0.0, rb=True)
    with gpu.Handle(N, dh, depth, flags=gpu.MG_FLAG_FAST_SMOOTHER) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        got = [h.vcycles(1) for _ in range(k)]
        assert_bit_equal(h.get_phi(), phi, "phi")
        for l in range(1, depth + 1):
            assert_bit_equal(h.level_array("R", l), s.R(l), f"R_{l}")
            assert_bit_equal(h.level_array("eps", l), s.eps(l), f"eps_{l}")
    assert got == pytest.approx(list(hist), rel=1e-12)
    # and the two forms of the update are the same smoother up to rounding
    assert (phi != exact).any() and rel_l2(phi, exact) < 1e-12


@pytest.mark.parametrize("n,depth", [(41, 2), (129, 3)])
def test_fast_converged_phi_matches_the_reference_algorithm(gpu, orc, n, depth):
    N, dh, rho = problem(orc, n)
    tol = 1e-6
    phi_ref = np.zeros(N)
    ok_ref, _, _, _ = orc.Solver(N, dh, depth).solve(phi_ref, rho, 10000, tol)   # lexicographic, exact update
    phi_twin = np.zeros(N)
    ok_twin, cyc_twin, l2_twin, _ = orc.Solver(N, dh, depth).solve(phi_twin, rho, 10000, tol, fast=True)
    assert ok_ref and ok_twin
    with gpu.Handle(N, dh, depth, flags=gpu.MG_FLAG_FAST_SMOOTHER) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        ok, cyc, l2 = h.solve(10000, tol)
        phi = h.get_phi()
    assert ok and l2 < tol and cyc == cyc_twin
    assert_bit_equal(phi, phi_twin, "phi vs the oracle's fast twin")
    assert rel_l2(phi, phi_ref) <= 1e-6
    # the norm the solve stopped on is the exact residual operator's
    _, s = orc.residual(phi, rho, orc.coeffs(dh, 0))
    assert np.sqrt(s / phi.size) == pytest.approx(l2, rel=1e-10)


def test_fast_smoother_on_slabs_equals_the_undivided_fast_handle(gpu):
    from multigrid_amd import problem as pb
    N = (129, 129, 129)
    dh = pb.main_extents_dh(N)
    g = rng(5)
    rho, phi0 = g.uniform(-1, 1, N) * 8.85e-9, g.uniform(-1, 1, N)
    out = []
    for devices in (None, [0, 0, 0]):
        with gpu.Handle(N, dh, 3, flags=gpu.MG_FLAG_FAST_SMOOTHER, devices=devices, coarse_cut=2 if devices else 0) as h:
            h.set_rho(rho)
            h.set_phi(phi0)
            out.append((h.vcycles(3), h.get_phi()))
    assert out[1][0] == pytest.approx(out[0][0], rel=1e-12)
    assert_bit_equal(out[1][1], out[0][1], "phi of the slabs, fast smoother")


def test_flag_leaves_the_fine_grid_sweeps_exact(gpu, orc):
    """mg_gs_solve sweeps the fine grid: with the flag it still runs the reference's expression."""
    N = (140, 133, 150)
    dh = orc.main_extents_dh(N)
    g = rng(11)
    rho, phi0 = g.uniform(-1, 1, N) * 8.85e-9, g.uniform(-1, 1, N)
    want = phi0.copy()
    orc.smooth(want, rho, N, orc.coeffs(dh, 0), sweeps=26, div_eps0=True, rb=True)
    with gpu.Handle(N, dh, 0, flags=gpu.MG_FLAG_FAST_SMOOTHER) as h:
        h.set_rho(rho)
        h.set_phi(phi0)
        h.gs_solve(26, 0.0)
        assert_bit_equal(h.get_phi(), want, "26 sweeps of the fine grid")


@pytest.mark.parametrize("shape", [(34, 34, 34), (66, 66, 66), (130, 130, 130), (21, 150, 139), (260, 131, 140)], ids=str)
def test_fast_sweeps_of_a_coarse_level_match_the_twin(gpu, orc, shape):
    """The short form kernel by kernel: a two-grid handle whose coarse level has the given eps shape
    (one node larger than R) and is swept seven times per cycle -- the LDS kernel, tiles, the
    marching sweeps and sweep pairs, depending on the shape.  eps_1 and phi after two cycles must
    equal the twin's."""
    n1 = tuple(a - 1 for a in shape)            # level-1 size
    N = tuple(2 * a + 1 for a in n1)            # fine size with N >> 1 == n1
    dh = orc.main_extents_dh(N)
    g = rng(3)
    rho, phi0 = g.uniform(-1, 1, N) * 8.85e-9, g.uniform(-1, 1, N)
    phi = phi0.copy()
    s = orc.Solver(N, dh, 1, nu=(1, 7))
    s.solve(phi, rho, 2, 0.0, fast=True)
    with gpu.Handle(N, dh, 1, nu=(1, 7), flags=gpu.MG_FLAG_FAST_SMOOTHER) as h:
        h.set_rho(rho)
        h.set_phi(phi0)
        h.vcycles(2)
        assert_bit_equal(h.level_array("eps", 1), s.eps(1), "eps_1 after 7 fast sweeps, twice")
        assert_bit_equal(h.get_phi(), phi, "phi")


def test_headline_size_fast_solve_meets_the_bar(gpu, orc):
    """513^3, five coarse grids, fast smoother, to 1e-6: the oracle's exact residual operator confirms
    the norm, and the potential equals the committed sample of the reference-order solve."""
    n = 513
    N, dh, rho = problem(orc, n)
    with gpu.Handle(N, dh, 5, flags=gpu.MG_FLAG_FAST_SMOOTHER) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        ok, cycles, l2 = h.solve(1000, 1e-6)
        phi = h.get_phi()
    assert ok and l2 < 1e-6 and 100 <= cycles <= 125
    _, s = orc.residual(phi, rho, orc.coeffs(dh, 0))
    assert np.sqrt(s / float(n) ** 3) == pytest.approx(l2, rel=1e-10)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "phi513_lex_sample.npz"))
    st = int(gold["stride"])
    rel = float(np.linalg.norm(phi[::st, ::st, ::st] - gold["phi_sample"]) / np.linalg.norm(gold["phi_sample"]))
    assert rel <= 1e-6, f"fast-smoother 513^3 potential differs from the reference-order solve by {rel:.3e}"
