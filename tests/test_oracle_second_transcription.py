"""A second, independent restatement of the reference's two-level solver, in plain Python loops,
against the C oracle -- TEST INFRASTRUCTURE ONLY.

oracle/mg_oracle.c restates the reference's hand-unrolled solvers as ONE depth-generic loop on flat
arrays.  The reference cannot be compiled in this image (its header needs <fftw3.h>), so nothing
reference-built pins that restatement; what can be done is to write the reference's text down a
second time, as literally as Python allows and without looking at the C -- one function per loop nest
of MultiGridSolverV2::solve() (cpp/PotentialSolver.cpp:2594-2879), every expression in the order it
is written there -- and require the two to agree bit for bit on small grids: potential, every work
array and the norm history.  A slip in either transcription (an index, a bound, a term order, an
association) shows up as a difference; the same slip made twice independently is what remains
uncovered.  Python floats are IEEE doubles without contraction, like the reference built by g++ for
x86-64.
"""
import math

import numpy as np
import pytest

from util import assert_bit_equal

EPS_0 = 8.85418782e-12  # cpp/World.h:14


class SecondV2:
    """MultiGridSolverV2 (cpp/PotentialSolver.h:197-214): members R_h, R_2h, R_4h, eps_h, eps_2h, eps_4h,
    zero-filled at construction and never cleared (cpp/Field.h:61-73)."""

    def __init__(self, n, dh):
        self.ni, self.nj, self.nk = n
        self.dh = dh
        ni, nj, nk = n
        self.R_h = np.zeros((ni, nj, nk))
        self.eps_h = np.zeros((ni, nj, nk))
        self.R_2h = np.zeros((ni // 2, nj // 2, nk // 2))
        self.eps_2h = np.zeros((ni // 2 + 1, nj // 2 + 1, nk // 2 + 1))
        self.R_4h = np.zeros((ni // 4, nj // 4, nk // 4))
        self.eps_4h = np.zeros((ni // 4 + 1, nj // 4 + 1, nk // 4 + 1))

    @staticmethod
    def _sor(u, rhs, n, cx, cy, cz, its, scale):
        # cpp/PotentialSolver.cpp:2637-2653 (fine: rhs/EPS_0), :2726-2741, :2783-2798 (coarse: rhs as it is)
        ni, nj, nk = n
        for _ in range(its):
            for i in range(1, ni - 1):
                for j in range(1, nj - 1):
                    for k in range(1, nk - 1):
                        b = rhs[i][j][k] / EPS_0 if scale else rhs[i][j][k]
                        g = (b +
                             cx * (u[i - 1][j][k] + u[i + 1][j][k]) +
                             cy * (u[i][j - 1][k] + u[i][j + 1][k]) +
                             cz * (u[i][j][k - 1] + u[i][j][k + 1])) / (2 * cx + 2 * cy + 2 * cz)
                        u[i][j][k] = u[i][j][k] + 1.4 * (g - u[i][j][k])

    @staticmethod
    def _restrict(Rf, Rc, n):
        # cpp/PotentialSolver.cpp:2686-2703 and :2706-2723
        ni, nj, nk = n
        for i in range(0, ni - 1, 2):
            for j in range(0, nj - 1, 2):
                for k in range(0, nk - 1, 2):
                    if i == 0 or j == 0 or k == 0:
                        Rc[i // 2][j // 2][k // 2] = Rf[i][j][k]
                    else:
                        Rc[i // 2][j // 2][k // 2] = (
                            Rf[i-1][j-1][k-1] + 2*Rf[i][j-1][k-1] + Rf[i+1][j-1][k-1] +
                            2*Rf[i-1][j][k-1] + 4*Rf[i][j][k-1] + 2*Rf[i+1][j][k-1] +
                            Rf[i-1][j+1][k-1] + 2*Rf[i][j+1][k-1] + Rf[i+1][j+1][k-1] +
                            2*Rf[i-1][j-1][k] + 4*Rf[i][j-1][k] + 2*Rf[i+1][j-1][k] +
                            4*Rf[i-1][j][k] + 8*Rf[i][j][k] + 4*Rf[i+1][j][k] +
                            2*Rf[i-1][j+1][k] + 4*Rf[i][j+1][k] + 2*Rf[i+1][j+1][k] +
                            Rf[i-1][j-1][k+1] + 2*Rf[i][j-1][k+1] + Rf[i+1][j-1][k+1] +
                            2*Rf[i-1][j][k+1] + 4*Rf[i][j][k+1] + 2*Rf[i+1][j][k+1] +
                            Rf[i-1][j+1][k+1] + 2*Rf[i][j+1][k+1] + Rf[i+1][j+1][k+1]) / 64.0

    @staticmethod
    def _interpolate(c, f, n):
        # cpp/PotentialSolver.cpp:2744-2780 and :2801-2837; n = the loop's node counts (world.ni/2 or world.ni ...)
        ni, nj, nk = n
        for i in range(0, ni - 1):
            for j in range(0, nj - 1):
                for k in range(0, nk - 1):
                    I, J, K = i // 2, j // 2, k // 2
                    if i == 0 or i == ni - 1 or j == 0 or j == nj - 1 or k == 0 or k == nk - 1:
                        f[i][j][k] = 0
                    elif i % 2 == 0 and j % 2 == 0 and k % 2 == 0:
                        f[i][j][k] = c[I][J][K]
                    elif i % 2 == 1 and j % 2 == 0 and k % 2 == 0:
                        f[i][j][k] = 0.5 * (c[I][J][K] + c[I + 1][J][K])
                    elif i % 2 == 0 and j % 2 == 1 and k % 2 == 0:
                        f[i][j][k] = 0.5 * (c[I][J][K] + c[I][J + 1][K])
                    elif i % 2 == 0 and j % 2 == 0 and k % 2 == 1:
                        f[i][j][k] = 0.5 * (c[I][J][K] + c[I][J][K + 1])
                    elif i % 2 == 1 and j % 2 == 1 and k % 2 == 0:
                        f[i][j][k] = 0.25 * (c[I][J][K] + c[I + 1][J][K] +
                                             c[I][J + 1][K] + c[I + 1][J + 1][K])
                    elif i % 2 == 1 and j % 2 == 0 and k % 2 == 1:
                        f[i][j][k] = 0.25 * (c[I][J][K] + c[I + 1][J][K] +
                                             c[I][J][K + 1] + c[I + 1][J][K + 1])
                    elif i % 2 == 0 and j % 2 == 1 and k % 2 == 1:
                        f[i][j][k] = 0.25 * (c[I][J][K] + c[I][J + 1][K] +
                                             c[I][J][K + 1] + c[I][J + 1][K + 1])
                    else:
                        f[i][j][k] = 0.125 * (c[I][J][K] + c[I + 1][J][K] +
                                              c[I][J + 1][K] + c[I + 1][J + 1][K] +
                                              c[I][J][K + 1] + c[I + 1][J][K + 1] +
                                              c[I][J + 1][K + 1] + c[I + 1][J + 1][K + 1])

    def solve(self, phi, rho, max_it, tol):
        """cpp/PotentialSolver.cpp:2594-2879.  Returns (converged, norm history)."""
        ni, nj, nk = self.ni, self.nj, self.nk
        dh = self.dh
        idx2, idy2, idz2 = 1.0 / (dh[0] * dh[0]), 1.0 / (dh[1] * dh[1]), 1.0 / (dh[2] * dh[2])
        idx2_2h, idy2_2h, idz2_2h = 1.0 / (4 * dh[0] * dh[0]), 1.0 / (4 * dh[1] * dh[1]), 1.0 / (4 * dh[2] * dh[2])
        idx2_4h, idy2_4h, idz2_4h = 1.0 / (16 * dh[0] * dh[0]), 1.0 / (16 * dh[1] * dh[1]), 1.0 / (16 * dh[2] * dh[2])
        hist = []
        converged = False
        for _ in range(max_it):
            self._sor(phi, rho, (ni, nj, nk), idx2, idy2, idz2, 3, True)               # 1)
            R_h = self.R_h
            for i in range(ni):                                                          # 2)
                for j in range(nj):
                    for k in range(nk):
                        if i == 0 or i == ni - 1 or j == 0 or j == nj - 1 or k == 0 or k == nk - 1:
                            R_h[i][j][k] = phi[i][j][k] - 0
                        else:
                            R_h[i][j][k] = (phi[i][j][k] * (2 * idx2 + 2 * idy2 + 2 * idz2) -
                                            rho[i][j][k] / EPS_0 -
                                            idx2 * (phi[i - 1][j][k] + phi[i + 1][j][k]) -
                                            idy2 * (phi[i][j - 1][k] + phi[i][j + 1][k]) -
                                            idz2 * (phi[i][j][k - 1] + phi[i][j][k + 1]))
            total = 0.0                                                                  # 2b)
            for i in range(ni):
                for j in range(nj):
                    for k in range(nk):
                        total += R_h[i][j][k] * R_h[i][j][k]
            L2 = math.sqrt(total / (ni * nj * nk))
            hist.append(L2)
            if L2 < tol:
                converged = True
                break
            self._restrict(R_h, self.R_2h, (ni, nj, nk))                                 # 3)
            self._restrict(self.R_2h, self.R_4h, (ni // 2, nj // 2, nk // 2))
            self._sor(self.eps_4h, self.R_4h, (ni // 4, nj // 4, nk // 4), idx2_4h, idy2_4h, idz2_4h, 50, False)  # 4)
            self._interpolate(self.eps_4h, self.eps_2h, (ni // 2, nj // 2, nk // 2))     # 5)
            self._sor(self.eps_2h, self.R_2h, (ni // 2, nj // 2, nk // 2), idx2_2h, idy2_2h, idz2_2h, 5, False)   # 4)
            self._interpolate(self.eps_2h, self.eps_h, (ni, nj, nk))                     # 5)
            for i in range(ni):                                                          # 6)
                for j in range(nj):
                    for k in range(nk):
                        phi[i][j][k] = phi[i][j][k] - self.eps_h[i][j][k]
        return converged, hist


class SecondVn(SecondV2):
    """The other classes are the same loop nests once more per level (MultiGridSolver :2390-2591,
    V3 :2882-3251, V4 :3254-3708, V5 :3711-4248): coarse size world.ni/2, /4, ... /32 (integer
    division), coefficients 1.0/(4*dh*dh), 16, 64, 256, 1024 (:3728-3750), R arrays of that size and eps
    arrays one node larger (cpp/PotentialSolver.h:176-178, :203, :222, :242, :261), and the sweep counts
    of each class (:2422-2425, :2629-2634, :2922-2929, :3298-3307, :3759-3770).  Written with the level
    as a loop variable; the loop bodies are SecondV2's."""

    SWEEPS = {1: (3, 50), 2: (3, 5, 50), 3: (3, 5, 5, 50), 4: (3, 5, 5, 5, 50), 5: (1, 5, 5, 5, 5, 50)}

    def __init__(self, n, dh, depth):
        self.n, self.dh, self.depth = n, dh, depth
        size = lambda l: tuple(a // (2 ** l) for a in n)
        self.R = [np.zeros(size(l)) for l in range(depth + 1)]
        self.eps = [np.zeros(n)] + [np.zeros(tuple(a + 1 for a in size(l))) for l in range(1, depth + 1)]

    def solve(self, phi, rho, max_it, tol):
        n, dh, L = self.n, self.dh, self.depth
        nu = self.SWEEPS[L]
        size = lambda l: tuple(a // (2 ** l) for a in n)
        coef = [tuple(1.0 / (dh[a] * dh[a]) if l == 0 else 1.0 / ((4 ** l) * dh[a] * dh[a]) for a in range(3)) for l in range(L + 1)]
        ni, nj, nk = n
        hist, converged = [], False
        for _ in range(max_it):
            self._sor(phi, rho, n, *coef[0], nu[0], True)
            cx, cy, cz = coef[0]
            R = self.R[0]
            for i in range(ni):
                for j in range(nj):
                    for k in range(nk):
                        if i == 0 or i == ni - 1 or j == 0 or j == nj - 1 or k == 0 or k == nk - 1:
                            R[i][j][k] = phi[i][j][k] - 0
                        else:
                            R[i][j][k] = (phi[i][j][k] * (2 * cx + 2 * cy + 2 * cz) -
                                          rho[i][j][k] / EPS_0 -
                                          cx * (phi[i - 1][j][k] + phi[i + 1][j][k]) -
                                          cy * (phi[i][j - 1][k] + phi[i][j + 1][k]) -
                                          cz * (phi[i][j][k - 1] + phi[i][j][k + 1]))
            total = 0.0
            for i in range(ni):
                for j in range(nj):
                    for k in range(nk):
                        total += R[i][j][k] * R[i][j][k]
            L2 = math.sqrt(total / (ni * nj * nk))
            hist.append(L2)
            if L2 < tol:
                converged = True
                break
            for l in range(L):
                self._restrict(self.R[l], self.R[l + 1], size(l))
            self._sor(self.eps[L], self.R[L], size(L), *coef[L], nu[L], False)
            for l in range(L - 1, -1, -1):
                self._interpolate(self.eps[l + 1], self.eps[l], size(l))
                if l > 0:
                    self._sor(self.eps[l], self.R[l], size(l), *coef[l], nu[l], False)
            for i in range(ni):
                for j in range(nj):
                    for k in range(nk):
                        phi[i][j][k] = phi[i][j][k] - self.eps[0][i][j][k]
        return converged, hist


@pytest.mark.parametrize("n,depth,cycles", [((9, 11, 13), 1, 3), ((25, 27, 29), 3, 2), ((49, 50, 51), 4, 2),
                                            ((97, 97, 100), 5, 1)], ids=str)
def test_c_oracle_equals_the_second_transcription_at_every_depth(orc, n, depth, cycles):
    """The reference's one-, three-, four- and five-level classes, with their own sweep counts, on
    non-cubic grids with even and odd node counts and unequal spacings."""
    g = np.random.default_rng(11 + sum(n))
    dh = (0.4 / (n[0] - 1), 0.37 / (n[1] - 1), 0.43 / (n[2] - 1))
    rho = g.uniform(-1, 1, n) * EPS_0 * 1e3
    phi0 = np.zeros(n)
    phi0[1:-1, 1:-1, 1:-1] = g.uniform(-1, 1, tuple(a - 2 for a in n))
    second = SecondVn(n, dh, depth)
    phi_b = phi0.copy()
    _, hist_b = second.solve(phi_b, rho, cycles, 0.0)
    first = orc.Solver(n, dh, depth)
    phi_a = phi0.copy()
    _, _, _, hist_a = first.solve(phi_a, rho.copy(), cycles, 0.0)
    assert [float(x) for x in hist_a] == hist_b
    assert_bit_equal(phi_a, phi_b, "phi")
    for l in range(depth + 1):
        assert_bit_equal(np.array(first.R(l)), second.R[l], f"R of level {l}")
        if l >= 1:
            assert_bit_equal(np.array(first.eps(l)), second.eps[l], f"eps of level {l}")


@pytest.mark.parametrize("n,dh,cycles", [((13, 13, 13), (0.4 / 12,) * 3, 4),
                                         ((17, 17, 17), (0.025, 0.025, 0.025), 3),
                                         ((13, 17, 21), (0.031, 0.023, 0.019), 3)], ids=str)
def test_c_oracle_equals_a_second_transcription_of_the_reference(orc, n, dh, cycles):
    """Random potential inside zero walls, random charge density: after a few iterations of the
    two-level solver the potential, all six work arrays and the norm of every iteration are the same
    bits in both restatements; the stopping test agrees too."""
    g = np.random.default_rng(5 + sum(n))
    rho = g.uniform(-1, 1, n) * EPS_0 * 1e3
    phi0 = np.zeros(n)
    phi0[1:-1, 1:-1, 1:-1] = g.uniform(-1, 1, tuple(a - 2 for a in n))

    second = SecondV2(n, dh)
    phi_b = phi0.copy()
    ok_b, hist_b = second.solve(phi_b, rho, cycles, 0.0)

    first = orc.Solver(n, dh, 2)
    phi_a = phi0.copy()
    ok_a, cyc_a, l2_a, hist_a = first.solve(phi_a, rho.copy(), cycles, 0.0)

    assert (ok_a, cyc_a) == (ok_b, cycles) == (False, cycles)
    assert [float(x) for x in hist_a] == hist_b and l2_a == hist_b[-1]
    assert_bit_equal(phi_a, phi_b, "phi")
    for name, a, b in [("R_h", first.R(0), second.R_h), ("R_2h", first.R(1), second.R_2h), ("R_4h", first.R(2), second.R_4h),
                       ("eps_2h", first.eps(1), second.eps_2h), ("eps_4h", first.eps(2), second.eps_4h)]:
        assert_bit_equal(np.array(a), b, name)
    # the stopping rule: a tolerance between the second and the third norm stops both in the third iteration
    tol = 0.5 * (hist_b[1] + hist_b[2]) if hist_b[2] < hist_b[1] else None
    if tol is not None:
        second2, first2 = SecondV2(n, dh), orc.Solver(n, dh, 2)
        pa, pb = phi0.copy(), phi0.copy()
        ok_b2, hist_b2 = second2.solve(pb, rho, 50, tol)
        ok_a2, cyc_a2, _, _ = first2.solve(pa, rho.copy(), 50, tol)
        assert ok_a2 and ok_b2 and cyc_a2 == len(hist_b2) == 3
        assert_bit_equal(pa, pb, "phi at the stopping iteration")


def test_field_gauss_seidel_and_seven_point_restriction_second_transcription(orc):
    """The rows next to the V-cycle, transcribed a second time the same way: E = -grad(phi)
    (cpp/PotentialSolver.cpp:418-458), GaussSeidelSolver::solve() (:235-291: one lexicographic SOR
    sweep per iteration, the interior residual norm tested when it % 25 == 0) and the "B" classes'
    restriction (:704-715)."""
    g = np.random.default_rng(77)
    n = (7, 9, 11)
    ni, nj, nk = n
    dh = (0.013, 0.017, 0.011)
    phi = g.uniform(-1, 1, n)
    # -- computeEF
    dx, dy, dz = dh
    ef = np.zeros(n + (3,))
    for i in range(ni):
        for j in range(nj):
            for k in range(nk):
                if i == 0:
                    ef[i][j][k][0] = -(-3 * phi[i][j][k] + 4 * phi[i + 1][j][k] - phi[i + 2][j][k]) / (2 * dx)
                elif i == ni - 1:
                    ef[i][j][k][0] = -(phi[i - 2][j][k] - 4 * phi[i - 1][j][k] + 3 * phi[i][j][k]) / (2 * dx)
                else:
                    ef[i][j][k][0] = -(phi[i + 1][j][k] - phi[i - 1][j][k]) / (2 * dx)
                if j == 0:
                    ef[i][j][k][1] = -(-3 * phi[i][j][k] + 4 * phi[i][j + 1][k] - phi[i][j + 2][k]) / (2 * dy)
                elif j == nj - 1:
                    ef[i][j][k][1] = -(phi[i][j - 2][k] - 4 * phi[i][j - 1][k] + 3 * phi[i][j][k]) / (2 * dy)
                else:
                    ef[i][j][k][1] = -(phi[i][j + 1][k] - phi[i][j - 1][k]) / (2 * dy)
                if k == 0:
                    ef[i][j][k][2] = -(-3 * phi[i][j][k] + 4 * phi[i][j][k + 1] - phi[i][j][k + 2]) / (2 * dz)
                elif k == nk - 1:
                    ef[i][j][k][2] = -(phi[i][j][k - 2] - 4 * phi[i][j][k - 1] + 3 * phi[i][j][k]) / (2 * dz)
                else:
                    ef[i][j][k][2] = -(phi[i][j][k + 1] - phi[i][j][k - 1]) / (2 * dz)
    assert_bit_equal(orc.compute_ef(phi.copy(), dh), ef, "E")

    # -- the seven-point restriction of the "B" classes
    R = g.uniform(-1, 1, (9, 12, 13))
    Rc = np.zeros(tuple(a // 2 for a in R.shape))
    for i in range(0, R.shape[0] - 1, 2):
        for j in range(0, R.shape[1] - 1, 2):
            for k in range(0, R.shape[2] - 1, 2):
                if i == 0 or j == 0 or k == 0:
                    Rc[i // 2][j // 2][k // 2] = R[i][j][k]
                else:
                    Rc[i // 2][j // 2][k // 2] = (R[i - 1][j][k] + 2 * R[i][j][k] + R[i + 1][j][k] +
                                                  R[i][j - 1][k] + 2 * R[i][j][k] + R[i][j + 1][k] +
                                                  R[i][j][k - 1] + 2 * R[i][j][k] + R[i][j][k + 1]) / 12.0
    assert_bit_equal(orc.restrict7(R.copy()), Rc, "seven-point restriction")

    # -- GaussSeidelSolver::solve()
    rho = g.uniform(-1, 1, n) * EPS_0 * 1e3
    u0 = np.zeros(n)
    u0[1:-1, 1:-1, 1:-1] = g.uniform(-1, 1, tuple(a - 2 for a in n))

    def gs(u, max_it, tol):
        idx2, idy2, idz2 = 1.0 / (dh[0] * dh[0]), 1.0 / (dh[1] * dh[1]), 1.0 / (dh[2] * dh[2])
        L2 = 0.0
        for it in range(max_it):
            for i in range(1, ni - 1):
                for j in range(1, nj - 1):
                    for k in range(1, nk - 1):
                        new = (rho[i][j][k] / EPS_0 +
                               idx2 * (u[i - 1][j][k] + u[i + 1][j][k]) +
                               idy2 * (u[i][j - 1][k] + u[i][j + 1][k]) +
                               idz2 * (u[i][j][k - 1] + u[i][j][k + 1])) / (2 * idx2 + 2 * idy2 + 2 * idz2)
                        u[i][j][k] = u[i][j][k] + 1.4 * (new - u[i][j][k])
            if it % 25 == 0:
                total = 0.0
                for i in range(1, ni - 1):
                    for j in range(1, nj - 1):
                        for k in range(1, nk - 1):
                            r = (-u[i][j][k] * (2 * idx2 + 2 * idy2 + 2 * idz2) +
                                 rho[i][j][k] / EPS_0 +
                                 idx2 * (u[i - 1][j][k] + u[i + 1][j][k]) +
                                 idy2 * (u[i][j - 1][k] + u[i][j + 1][k]) +
                                 idz2 * (u[i][j][k - 1] + u[i][j][k + 1]))
                            total += r * r
                L2 = math.sqrt(total / (ni * nj * nk))
                if L2 < tol:
                    return True, it, L2
        return False, max_it, L2

    for max_it, tol in ((60, 0.0), (400, 1e-3)):
        ub, ua = u0.copy(), u0.copy()
        ok_b, it_b, l2_b = gs(ub, max_it, tol)
        ok_a, its_a, l2_a = orc.gs_solve(ua, rho.copy(), dh, max_it, tol)
        assert ok_a == ok_b and l2_a == l2_b
        assert_bit_equal(ua, ub, "phi after Gauss-Seidel")
        if ok_b:  # the sweep count the oracle reports: iterations run, the stopping one included
            assert its_a == it_b + 1
