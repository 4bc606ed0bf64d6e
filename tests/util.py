"""Shared helpers of the test-suite (inputs, comparisons)."""
import numpy as np

EPS_0 = 8.85418782e-12


def rng(seed=12345):
    return np.random.default_rng(seed)


def phi_like(shape, g, zero_faces=False):
    a = g.uniform(-1.0, 1.0, size=shape) * 1e3
    if zero_faces:
        a[0], a[-1] = 0, 0
        a[:, 0], a[:, -1] = 0, 0
        a[:, :, 0], a[:, :, -1] = 0, 0
    return np.ascontiguousarray(a)


def rho_like(shape, g):
    return np.ascontiguousarray(g.uniform(-1.0, 1.0, size=shape) * EPS_0 * 1e3)


def assert_bit_equal(got, want, what=""):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, what
    same = got.view(np.uint64) == want.view(np.uint64)
    if not same.all():
        bad = np.argwhere(~same)
        idx = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {got.size} elements differ bitwise; first at {idx}: "
                             f"got {got[idx]!r} want {want[idx]!r}")


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
