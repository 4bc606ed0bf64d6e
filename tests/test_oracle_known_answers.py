"""Pins the CPU oracle's lexicographic solver to the residual-norm histories that
were recorded from the compiled reference (tests/golden/survey_known_answers.json).

The reference holds no tests or fixtures for this path and cannot be compiled in
this image (it needs <fftw3.h>), so these recorded values are the only
reference-derived numbers available; they agree to the 7 digits recorded."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "survey_known_answers.json")))["cases"]


def _run(orc, n, depth, max_it):
    N = (n, n, n)
    dh = orc.main_extents_dh(N)
    rho = orc.synthetic_rho(N)
    phi = np.zeros(N)
    s = orc.Solver(N, dh, depth)
    return s.solve(phi, rho, max_it, 1e-6)


# full solves where they take seconds; the first cycles only at 257^3 / 513^3
@pytest.mark.parametrize("case,max_it", [(KA[0], 200), (KA[1], 200), (KA[2], 4), (KA[3], 4)],
                         ids=["41_V2", "129_V3", "257_V5_first4", "513_V5_first4"])
def test_l2_history_matches_reference_record(orc, case, max_it):
    ok, cycles, l2, hist = _run(orc, case["n"], case["depth"], max_it)
    checked = 0
    for it, want in case["l2"].items():
        it = int(it)
        if it < len(hist):
            assert hist[it] == pytest.approx(want, rel=5e-7), f"iteration {it}"
            checked += 1
    assert checked >= 3
    if case["converged_at"] is not None and max_it > case["converged_at"]:
        assert ok
        # the reference breaks in iteration index `converged_at`
        assert cycles == case["converged_at"] + 1
        assert l2 < 1e-6


def test_compute_ef_is_exact_for_quadratics(orc):
    """Central and 3-point one-sided differences are exact for a quadratic potential, so
    E = -grad(phi) must come out analytically on every node, faces included
    (cpp/PotentialSolver.cpp:418-458)."""
    import numpy as np
    n = (9, 12, 7)
    dh = (0.05, 0.03, 0.07)
    x = np.arange(n[0])[:, None, None] * dh[0]
    y = np.arange(n[1])[None, :, None] * dh[1]
    z = np.arange(n[2])[None, None, :] * dh[2]
    phi = np.ascontiguousarray(3 * x * x - 2 * y * y + 0.5 * z * z + x * y - 4 * z + 1 + 0 * (x + y + z))
    ef = orc.compute_ef(phi, dh)
    want = np.stack(np.broadcast_arrays(-(6 * x + y), -(-4 * y + x), -(z - 4 + 0 * x)), axis=-1)
    assert np.allclose(ef, want, rtol=0, atol=1e-11)


def test_gs_loop_stops_on_a_tested_iteration_and_agrees_with_multigrid(orc):
    """The reference's GaussSeidelSolver tests convergence only when it % 25 == 0
    (cpp/PotentialSolver.cpp:268); its fixed point is the V-cycle's fixed point."""
    import numpy as np
    n = (21, 21, 21)
    dh, rho = orc.main_extents_dh(n), orc.synthetic_rho(n)
    gs = np.zeros(n)
    ok, its, l2 = orc.gs_solve(gs, rho, dh, 20000, 1e-7)
    assert ok and l2 < 1e-7 and its % 25 == 1
    mg = np.zeros(n)
    assert orc.Solver(n, dh, 1).solve(mg, rho, 500, 1e-7)[0]
    assert np.linalg.norm(gs - mg) / np.linalg.norm(mg) < 1e-7
    rb = np.zeros(n)
    ok_rb, its_rb, _ = orc.gs_solve(rb, rho, dh, 20000, 1e-7, rb=True)
    assert ok_rb and np.linalg.norm(rb - gs) / np.linalg.norm(gs) < 1e-7


def test_restrict7_weights_and_b_variant_fixed_point(orc):
    """(1+2+1)*3/12 = 1: a constant field restricts to itself; injection where an index is 0; and
    the B cycle (cpp/PotentialSolver.cpp:620-816) converges to the fixed point of the standard one."""
    import numpy as np
    Rf = np.full((9, 11, 13), 2.5)
    Rc = orc.restrict7(Rf)
    assert np.array_equal(Rc, np.full((4, 5, 6), 2.5))
    g = np.random.default_rng(5)
    Rf = g.uniform(-1, 1, size=(9, 9, 9))
    Rc = orc.restrict7(Rf)
    assert Rc[0, 2, 3] == Rf[0, 4, 6] and Rc[2, 0, 1] == Rf[4, 0, 2]
    c = Rf[4, 4, 4]
    want = (Rf[3, 4, 4] + 2 * c + Rf[5, 4, 4] + Rf[4, 3, 4] + 2 * c + Rf[4, 5, 4] + Rf[4, 4, 3] + 2 * c + Rf[4, 4, 5]) / 12.0
    assert Rc[2, 2, 2] == want
    n = (33, 33, 33)
    dh, rho = orc.main_extents_dh(n), orc.synthetic_rho(n)
    a, b = np.zeros(n), np.zeros(n)
    assert orc.Solver(n, dh, 2).solve(a, rho, 2000, 1e-7)[0]
    assert orc.Solver(n, dh, 2, variant_b=True).solve(b, rho, 2000, 1e-7)[0]
    assert np.linalg.norm(a - b) / np.linalg.norm(a) < 1e-7
    assert orc.schedule(5, variant_b=True) == (1, 2, 3, 4, 5, 50)
