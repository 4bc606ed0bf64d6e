"""Deep halos on the coarse distributed levels: same bits as the undivided handle."""
import numpy as np
import pytest

from test_gpu_dist_native import run_ranks
from util import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,n,depth,cut,deep", [(2, 257, 5, 3, True), (3, 257, 5, 3, True), (4, 257, 5, 3, False),
                                                    (4, 513, 5, 3, True), (2, 129, 4, 2, True), (2, 513, 5, 4, True)])
@pytest.mark.parametrize("hold_back", [False, True], ids=["early", "late"])
def test_deep(gpu, world, n, depth, cut, deep, hold_back):
    from multigrid_amd import problem
    N = (n, n, n)

    def body(d, r):
        assert d.deep_halos == deep, (d.deep_halos, deep)
        d.vcycles(3)
        lo, phi = d.get_phi_owned()
        return lo, phi

    res = run_ranks(gpu, world, n, depth, cut, body, hold_back)
    with gpu.Handle(N, problem.main_extents_dh(N), depth, device=0) as h:
        h.set_rho(problem.synthetic_rho(N))
        h.set_phi(np.zeros(N))
        h.vcycles(3)
        want = h.get_phi()
    got = np.full(N, np.nan)
    for lo, part in res:
        got[lo:lo + part.shape[0]] = part
    assert_bit_equal(got, want, "phi assembled from the slabs")
