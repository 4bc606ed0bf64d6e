"""The operators on shapes drawn at random (fixed seed): window, chunk and pair boundaries of the
marching kernels fall somewhere else for every shape, and so do the switches between kernel
variants (one workgroup in LDS, colour launches, single-pass sweep, sweep pairs, 32- and 64-row
windows).  Same bar as the per-operator tests: the oracle's bits."""
import os

import numpy as np
import pytest

from util import assert_bit_equal, phi_like, rho_like, rng

pytestmark = pytest.mark.gpu


def _shapes():
    # MG_FUZZ_SEED / MG_FUZZ_MULT draw other and more shapes for an occasional longer run
    g = np.random.default_rng(int(os.environ.get("MG_FUZZ_SEED", "20261111")))
    mult = int(os.environ.get("MG_FUZZ_MULT", "1"))
    out = []
    for lo, hi, count in ((3, 40, 4), (40, 140, 5), (120, 300, 5)):
        for _ in range(count * mult):
            out.append(tuple(int(x) for x in g.integers(lo, hi, size=3)))
    # one long axis each way round (few planes and many rows, many planes and few columns)
    out += [(5, 290, 131), (300, 9, 140), (131, 270, 6)]
    return out


SHAPES = _shapes()


def dh_for(shape):
    return tuple(0.4 / (s - 1) for s in shape)


@pytest.mark.parametrize("shape", SHAPES, ids=str)
def test_sweeps_on_random_shapes(gpu, orc, shape):
    g = rng(sum(shape))
    sweeps = 1 + sum(shape) % 5
    u, rho = phi_like(shape, g), rho_like(shape, g)
    c = orc.coeffs(dh_for(shape), 0)
    want = orc.smooth(u.copy(), rho, shape, c, 1.4, sweeps, div_eps0=True, rb=True)
    got = gpu.op_smooth_rb(u.copy(), rho, shape, c, 1.4, sweeps, rhs_is_rho=True)
    assert_bit_equal(got, want, f"{sweeps} sweeps, fine form")
    if min(shape) >= 4:
        n = tuple(a - 1 for a in shape)  # coarse form: the array is one node larger than the level
        R = phi_like(n, g)
        c = orc.coeffs(dh_for(tuple(2 * a + 1 for a in n)), 1)
        want = orc.smooth(u.copy(), R, n, c, 1.4, sweeps, div_eps0=False, rb=True)
        got = gpu.op_smooth_rb(u.copy(), R, n, c, 1.4, sweeps, rhs_is_rho=False)
        assert_bit_equal(got, want, f"{sweeps} sweeps, coarse form")


@pytest.mark.parametrize("shape", SHAPES, ids=str)
def test_residual_restriction_and_field_on_random_shapes(gpu, orc, shape):
    g = rng(7 + sum(shape))
    phi, rho = phi_like(shape, g), rho_like(shape, g)
    dh = dh_for(shape)
    want_R, want_sum = orc.residual(phi, rho, orc.coeffs(dh, 0))
    nc = tuple(a >> 1 for a in shape)
    keep = phi_like(nc, g)  # entries the restriction never writes must stay
    want_c = orc.restrict27(want_R, keep.copy())
    _, got_c, got_sum = gpu.op_residual_restrict(phi, rho, dh, keep.copy())
    assert_bit_equal(got_c, want_c, "residual + restriction")
    assert got_sum == pytest.approx(want_sum, rel=1e-11)
    assert_bit_equal(gpu.op_compute_ef(phi, dh), orc.compute_ef(phi, dh), "E field")


@pytest.mark.parametrize("shape", SHAPES[4:], ids=str)
def test_prolongation_and_correction_on_random_shapes(gpu, orc, shape):
    g = rng(11 + sum(shape))
    ec = phi_like(tuple((a >> 1) + 1 for a in shape), g)
    phi = phi_like(shape, g)
    eps_h = np.zeros(shape)
    orc.prolong(ec, eps_h, shape)
    want = orc.correct(phi.copy(), eps_h)
    assert_bit_equal(gpu.op_prolong_correct(ec, phi.copy()), want, "prolongation + correction")


def _cycle_cases():
    g = np.random.default_rng(int(os.environ.get("MG_FUZZ_SEED", "20261111")) + 1)
    out = []
    for _ in range(8 * int(os.environ.get("MG_FUZZ_MULT", "1"))):
        hi = int(g.choice([40, 90, 160, 280]))
        n = tuple(int(x) for x in g.integers(12, hi, size=3))
        deepest = 0
        while (min(n) >> (deepest + 1)) >= 3:
            deepest += 1
        out.append((n, int(g.integers(1, min(deepest, 6) + 1)), ("plain", "keep_fine_residual", "b_variant")[int(g.choice([0, 0, 1, 2]))],
                    int(g.integers(1, 4))))
    return out


@pytest.mark.parametrize("n,depth,mode,cycles", _cycle_cases(), ids=str)
def test_whole_cycles_on_random_shapes(gpu, orc, n, depth, mode, cycles):
    """The whole V-cycle (every level's kernel choice follows from the shape) on non-cubic grids of
    random size and depth, in the default form, with the fine residual kept, and as the "B" variant:
    phi after a few iterations equals the oracle's red-black twin bit for bit."""
    g = rng(13 + sum(n))
    dh = dh_for(n)
    rho, phi0 = g.uniform(-1, 1, n) * 8.85e-9, g.uniform(-1, 1, n)
    flags = gpu.MG_FLAG_KEEP_FINE_RESIDUAL if mode == "keep_fine_residual" else 0
    with gpu.Handle(n, dh, depth, flags=flags, variant_b=mode == "b_variant") as h:
        h.set_rho(rho)
        h.set_phi(phi0)
        l2 = h.vcycles(cycles)
        phi = h.get_phi()
    want = phi0.copy()
    ref = orc.Solver(n, dh, depth, variant_b=mode == "b_variant").solve(want, rho, cycles, 0.0, rb=True)
    assert l2 == pytest.approx(ref[2], rel=1e-11)
    assert_bit_equal(phi, want, "phi")
