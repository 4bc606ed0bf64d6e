"""Analytic known answers for the CPU oracle's operators, derived from the reference's loop
definitions (cpp/PotentialSolver.cpp:3792-3805 residual, :3822-3919 restriction, :4170-4206
prolongation, :3773-3789 smoother, :3728-3750 coefficients).  The reference holds no fixtures for
these operators and cannot be compiled in this image; these identities pin what each operator
must do independently of any implementation."""
import numpy as np
import pytest


def test_level_coefficients_are_exact_quarters(orc):
    dh = (0.4 / 40, 0.4 / 40, 0.4 / 40)
    c0 = orc.coeffs(dh, 0)
    assert c0[0] == 1.0 / (dh[0] * dh[0])
    assert c0[3] == 2 * c0[0] + 2 * c0[1] + 2 * c0[2]
    for l in range(1, 6):
        c = orc.coeffs(dh, l)
        assert c[:3] == tuple(a / 4.0 ** l for a in c0[:3])  # scaling by a power of two is exact


@pytest.mark.parametrize("nf", [(17, 17, 17), (16, 16, 16), (9, 12, 15)], ids=str)
def test_restriction_weights(orc, nf):
    """Weights (1,2,1)^3 / 64: a constant stays a constant; a unit impulse spreads by its
    position's parity; index-0 planes are injected."""
    Rc = orc.restrict27(np.ones(nf))
    assert (Rc[:(nf[0] - 1 + 1) // 2, :(nf[1] - 1 + 1) // 2, :(nf[2] - 1 + 1) // 2] == 1.0).all()
    for pos, want in [((4, 4, 4), {(2, 2, 2): 8 / 64}),                       # coarse node itself
                      ((5, 4, 4), {(2, 2, 2): 4 / 64, (3, 2, 2): 4 / 64}),    # odd along i
                      ((5, 5, 4), {(2, 2, 2): 2 / 64, (3, 2, 2): 2 / 64, (2, 3, 2): 2 / 64, (3, 3, 2): 2 / 64}),
                      ((5, 5, 5), {(I, J, K): 1 / 64 for I in (2, 3) for J in (2, 3) for K in (2, 3)})]:
        Rf = np.zeros(nf)
        Rf[pos] = 1.0
        Rc = orc.restrict27(Rf)
        nz = {tuple(int(v) for v in idx): Rc[tuple(idx)] for idx in np.argwhere(Rc != 0)}
        assert nz == want, pos
    # injection where an index is 0: the coarse value is the fine value at (2I, 2J, 2K), no averaging
    g = np.random.default_rng(5)
    Rf = g.standard_normal(nf)
    Rc = orc.restrict27(Rf)
    cI, cJ, cK = ((nf[0] - 1 + 1) // 2, (nf[1] - 1 + 1) // 2, (nf[2] - 1 + 1) // 2)
    assert (Rc[0, :cJ, :cK] == Rf[0, 0:2 * cJ:2, 0:2 * cK:2]).all()
    assert (Rc[:cI, 0, :cK] == Rf[0:2 * cI:2, 0, 0:2 * cK:2]).all()
    assert (Rc[:cI, :cJ, 0] == Rf[0:2 * cI:2, 0:2 * cJ:2, 0]).all()


@pytest.mark.parametrize("n", [(17, 17, 17), (16, 16, 16), (9, 12, 15)], ids=str)
def test_prolongation_is_trilinear_and_zero_on_index_0(orc, n):
    cn = tuple((a >> 1) + 1 for a in n)
    I, J, K = np.meshgrid(*(np.arange(a, dtype=float) for a in cn), indexing="ij")
    ec = 3.0 * I - 2.0 * J + 0.5 * K + 7.0          # linear: trilinear interpolation reproduces it exactly
    ef = np.full(n, np.nan)
    orc.prolong(ec, ef, n)
    i, j, k = np.meshgrid(*(np.arange(a, dtype=float) for a in n), indexing="ij")
    want = 3.0 * i / 2 - 2.0 * j / 2 + 0.5 * k / 2 + 7.0
    inside = (i >= 1) & (j >= 1) & (k >= 1) & (i <= n[0] - 2) & (j <= n[1] - 2) & (k <= n[2] - 2)
    assert (ef[inside] == want[inside]).all()       # halves and quarters of small integers: exact
    zero = ((i == 0) | (j == 0) | (k == 0)) & (i <= n[0] - 2) & (j <= n[1] - 2) & (k <= n[2] - 2)
    assert (ef[zero] == 0.0).all()
    untouched = (i > n[0] - 2) | (j > n[1] - 2) | (k > n[2] - 2)
    assert np.isnan(ef[untouched]).all()            # the loops stop at n-2


def test_residual_of_a_discrete_solution_and_faces(orc):
    """phi = x^2 + 2 y^2 - z^2 in node coordinates: its 7-point Laplacian is exactly 2 + 4 - 2 = 4,
    so with rho/EPS_0 = -4 the interior residual A*phi - b vanishes (to rounding of the large
    terms); on the six faces the residual is phi itself."""
    n = (13, 11, 15)
    dh = (0.01, 0.02, 0.04)
    i, j, k = np.meshgrid(*(np.arange(a, dtype=float) for a in n), indexing="ij")
    x, y, z = i * dh[0], j * dh[1], k * dh[2]
    phi = x * x + 2 * y * y - z * z
    rho = np.full(n, -4.0 * orc.EPS_0)
    c = orc.coeffs(dh, 0)
    R, s = orc.residual(phi, rho, c)
    scale = np.abs(phi).max() * c[3]
    assert np.abs(R[1:-1, 1:-1, 1:-1]).max() <= 1e-13 * scale
    for f in (np.s_[0], np.s_[-1], np.s_[:, 0], np.s_[:, -1], np.s_[:, :, 0], np.s_[:, :, -1]):
        assert (R[f] == phi[f]).all()
    assert s == pytest.approx(float((R * R).sum()), rel=1e-12)


@pytest.mark.parametrize("rb", [False, True], ids=["lexicographic", "red_black"])
def test_smoother_keeps_a_discrete_solution_and_never_writes_faces(orc, rb):
    n = (13, 11, 15)
    dh = (0.01, 0.02, 0.04)
    i, j, k = np.meshgrid(*(np.arange(a, dtype=float) for a in n), indexing="ij")
    phi = (i * dh[0]) ** 2 + 2 * (j * dh[1]) ** 2 - (k * dh[2]) ** 2
    b = np.full(n, -4.0)                      # rhs already divided by EPS_0
    c = orc.coeffs(dh, 0)
    u = phi.copy()
    orc.smooth(u, b, n, c, 1.4, 3, rb=rb)
    assert np.abs(u - phi).max() <= 1e-12 * np.abs(phi).max()
    # faces hold the Dirichlet values and are never written, whatever the interior does
    u = np.random.default_rng(2).standard_normal(n)
    faces = u.copy()
    orc.smooth(u, b, n, c, 1.4, 2, rb=rb)
    faces[1:-1, 1:-1, 1:-1] = u[1:-1, 1:-1, 1:-1]
    assert (faces == u).all()


def test_one_point_update_formula(orc):
    """u += 1.4 * (g - u), g = (b + cx*(W+E) + cy*(S+N) + cz*(D+U)) / d at the single interior
    point of a 3^3 grid."""
    n = (3, 3, 3)
    g = np.random.default_rng(9)
    u = g.standard_normal(n)
    b = g.standard_normal(n)
    c = orc.coeffs((0.1, 0.2, 0.3), 0)
    want = u[1, 1, 1] + 1.4 * ((b[1, 1, 1] + c[0] * (u[0, 1, 1] + u[2, 1, 1]) + c[1] * (u[1, 0, 1] + u[1, 2, 1]) +
                                c[2] * (u[1, 1, 0] + u[1, 1, 2])) / c[3] - u[1, 1, 1])
    got = orc.smooth(u.copy(), b, n, c, 1.4, 1)
    assert got[1, 1, 1] == want
    assert (np.delete(got.ravel(), 13) == np.delete(u.ravel(), 13)).all()


def test_correction_subtracts_everywhere(orc):
    g = np.random.default_rng(4)
    phi, eps = g.standard_normal((5, 6, 7)), g.standard_normal((5, 6, 7))
    assert (orc.correct(phi.copy(), eps) == phi - eps).all()
