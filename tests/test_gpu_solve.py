"""Whole-solve parity: the HIP cycle driver against the oracle's red-black twin
(bit-exact state) and against the oracle's lexicographic solver, which is the
reference's algorithm (converged phi within 1e-6 relative L2 at the same
residual tolerance -- exact iterate agreement is impossible, the orderings
differ)."""
import numpy as np
import pytest

from util import assert_bit_equal, rel_l2

pytestmark = pytest.mark.gpu


def problem(orc, n):
    N = (n, n, n)
    return N, orc.main_extents_dh(N), orc.synthetic_rho(N)


@pytest.mark.parametrize("keep_rf", [False, True], ids=["default", "keep_fine_residual"])
@pytest.mark.parametrize("n,depth", [(41, 1), (41, 2), (65, 3), (129, 3), (129, 4), (129, 5), (257, 5)])
def test_cycles_match_rb_oracle_bit_for_bit(gpu, orc, n, depth, keep_rf):
    """After k iterations phi, R_l and eps_l equal the oracle's red-black twin.
    By default R_0 is never stored (fused residual+restriction); with
    MG_FLAG_KEEP_FINE_RESIDUAL the unfused kernels run and R_0 is compared too."""
    N, dh, rho = problem(orc, n)
    k = 3
    phi = np.zeros(N)
    s = orc.Solver(N, dh, depth)
    ok, cyc, l2, hist = s.solve(phi, rho, k, 0.0, rb=True)
    with gpu.Handle(N, dh, depth, flags=gpu.MG_FLAG_KEEP_FINE_RESIDUAL if keep_rf else 0) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        got_l2 = [h.vcycles(1) for _ in range(k)]
        assert_bit_equal(h.get_phi(), phi, "phi")
        if not keep_rf:
            with pytest.raises(gpu.MgError):
                h.level_array("R", 0)
        for l in range(0 if keep_rf else 1, depth + 1):
            assert_bit_equal(h.level_array("R", l), s.R(l), f"R_{l}")
        for l in range(1, depth + 1):
            assert_bit_equal(h.level_array("eps", l), s.eps(l), f"eps_{l}")
    assert got_l2 == pytest.approx(list(hist), rel=1e-12)


@pytest.mark.parametrize("n,depth", [(41, 2), (129, 3)])
def test_converged_phi_matches_reference_algorithm(gpu, orc, n, depth):
    N, dh, rho = problem(orc, n)
    tol = 1e-6
    phi_ref = np.zeros(N)
    ok_ref, cyc_ref, l2_ref, _ = orc.Solver(N, dh, depth).solve(phi_ref, rho, 10000, tol)
    phi_rb = np.zeros(N)
    ok_rb, cyc_rb, l2_rb, _ = orc.Solver(N, dh, depth).solve(phi_rb, rho, 10000, tol, rb=True)
    assert ok_ref and ok_rb
    with gpu.Handle(N, dh, depth) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        ok, cyc, l2 = h.solve(10000, tol)
        phi = h.get_phi()
    assert ok and l2 < tol
    assert cyc == cyc_rb                      # same stopping iteration as the CPU twin
    assert_bit_equal(phi, phi_rb, "phi vs red-black oracle")
    assert rel_l2(phi, phi_ref) <= 1e-6       # vs the reference's lexicographic solve
    assert l2 == pytest.approx(l2_rb, rel=1e-10)


def test_solve_stops_before_correction_like_the_reference(gpu, orc):
    """The returned phi is the one whose residual passed: recomputing the residual
    of the returned phi reproduces the reported norm."""
    N, dh, rho = problem(orc, 65)
    with gpu.Handle(N, dh, 3) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        ok, cyc, l2 = h.solve(10000, 1e-4)
        phi = h.get_phi()
    assert ok
    R, s = gpu.op_residual(phi, rho, dh)
    assert np.sqrt(s / phi.size) == pytest.approx(l2, rel=1e-12)
    assert np.sqrt(s / phi.size) < 1e-4


def test_not_converged_is_reported_not_raised(gpu, orc):
    N, dh, rho = problem(orc, 41)
    with gpu.Handle(N, dh, 2) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        ok, cyc, l2 = h.solve(2, 1e-12)
    assert not ok and cyc == 2 and l2 > 1.0


def test_warm_start_state_persists_between_solves(gpu, orc):
    """eps_l are never re-zeroed (reference members): two solve(1) calls equal one solve(2)."""
    N, dh, rho = problem(orc, 41)
    with gpu.Handle(N, dh, 2) as a, gpu.Handle(N, dh, 2) as b:
        for h in (a, b):
            h.set_rho(rho)
            h.set_phi(np.zeros(N))
        a.solve(1, 0.0)
        a.solve(1, 0.0)
        b.solve(2, 0.0)
        assert_bit_equal(a.get_phi(), b.get_phi(), "phi")
        a.reset_state()
        assert not a.level_array("eps", 2).any()


def test_zero_rho_converges_in_one_check(gpu, orc):
    """The reference driver's first solve() runs on rho == 0 (cpp/Main.cpp:124)."""
    N = (41, 41, 41)
    dh = orc.main_extents_dh(N)
    with gpu.Handle(N, dh, 2) as h:
        h.set_rho(np.zeros(N))
        h.set_phi(np.zeros(N))
        ok, cyc, l2 = h.solve(10000, 1e-4)
        assert ok and cyc == 1 and l2 == 0.0
        assert not h.get_phi().any()


def test_non_cubic_grid(gpu, orc):
    N = (33, 41, 65)
    dh = (0.01, 0.02, 0.005)
    rho = orc.synthetic_rho(N)
    phi = np.zeros(N)
    s = orc.Solver(N, dh, 2)
    s.solve(phi, rho, 4, 0.0, rb=True)
    with gpu.Handle(N, dh, 2) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        h.vcycles(4)
        assert_bit_equal(h.get_phi(), phi, "phi non-cubic")


@pytest.mark.parametrize("N,depth", [((67, 261, 133), 2), ((130, 140, 300), 3), ((40, 520, 270), 2)], ids=str)
def test_non_cubic_grids_on_the_marching_kernels(gpu, orc, N, depth):
    """Ragged, non-cubic sizes above 128 nodes: single-pass sweep (32- and 64-row windows), fused
    residual+restriction and on-load correction/prolongation with windows cut by the array edges."""
    dh = (0.01, 0.02, 0.005)
    rho = orc.synthetic_rho(N)
    phi = np.zeros(N)
    s = orc.Solver(N, dh, depth)
    ok, cyc, l2, hist = s.solve(phi, rho, 3, 0.0, rb=True)
    with gpu.Handle(N, dh, depth) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        l2_gpu = h.vcycles(3)
        assert_bit_equal(h.get_phi(), phi, "phi non-cubic")
        for l in range(1, depth + 1):
            assert_bit_equal(h.level_array("eps", l), s.eps(l), f"eps_{l}")
    assert l2_gpu == pytest.approx(hist[-1], rel=1e-11)


def test_state_errors(gpu):
    with gpu.Handle((41, 41, 41), (0.01,) * 3, 2) as h:
        with pytest.raises(gpu.MgError):
            h.solve(1, 1e-4)  # no rho/phi yet
    with pytest.raises(gpu.MgError):
        gpu.Handle((41, 41, 41), (0.01,) * 3, 5)  # 41 >> 5 == 1 node: too deep


def test_headline_size_first_cycles(gpu, orc):
    """513^3 V5: two iterations bit-identical to the oracle's red-black twin."""
    N, dh, rho = problem(orc, 513)
    phi = np.zeros(N)
    s = orc.Solver(N, dh, 5)
    ok, cyc, l2, hist = s.solve(phi, rho, 2, 0.0, rb=True)
    with gpu.Handle(N, dh, 5) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        got = [h.vcycles(1), h.vcycles(1)]
        got_phi = h.get_phi()
    assert_bit_equal(got_phi, phi, "phi 513^3")
    assert got == pytest.approx(list(hist), rel=1e-12)


def test_device_memory_estimate_matches_what_a_handle_allocates(gpu, orc):
    """mg_device_bytes against the free device memory before and after mg_create (allocations are
    rounded up to the driver's granularity, so the handle may take a little more, never less)."""
    import torch
    N = (257, 257, 257)
    dh = orc.main_extents_dh(N)
    want = gpu.device_bytes(N, dh, 5, flags=gpu.MG_FLAG_RESERVE_EF)
    with gpu.Handle((41, 41, 41), orc.main_extents_dh((41, 41, 41)), 2) as h:  # the runtime's own first-use allocations
        h.set_phi(np.zeros((41, 41, 41)))
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info(0)[0]
    with gpu.Handle(N, dh, 5, flags=gpu.MG_FLAG_RESERVE_EF):
        taken = before - torch.cuda.mem_get_info(0)[0]
    assert want <= taken <= want * 1.03 + (64 << 20)


def test_out_of_device_memory_says_what_the_configuration_needs(gpu, orc):
    """2049^3 with seven coarse grids, the fine residual kept and the field's array reserved needs
    more than one GPU has: mg_create fails with MG_ERR_NOMEM, names the need, and leaves the device
    usable."""
    N = (2049, 2049, 2049)
    dh = orc.main_extents_dh(N)
    flags = gpu.MG_FLAG_KEEP_FINE_RESIDUAL | gpu.MG_FLAG_RESERVE_EF
    need = gpu.device_bytes(N, dh, 7, flags=flags)
    assert need > 500e9
    with pytest.raises(gpu.MgError) as e:
        gpu.Handle(N, dh, 7, flags=flags)
    assert e.value.code == -3 and f"{need / 1e9:.2f} GB" in str(e.value)
    with gpu.Handle((41, 41, 41), orc.main_extents_dh((41, 41, 41)), 2) as h:  # nothing leaked, nothing sticky
        h.set_rho(orc.synthetic_rho((41, 41, 41)))
        h.set_phi(np.zeros((41, 41, 41)))
        assert np.isfinite(h.vcycles(1))


def test_handles_give_back_what_they_took(gpu, orc):
    """Creating, using and destroying handles (single GPU, several slabs on one GPU, one rank driven
    directly) leaves the free device memory where it was."""
    import torch
    from multigrid_amd import problem
    N = (65, 65, 65)
    dh = orc.main_extents_dh(N)
    rho = orc.synthetic_rho(N)

    def one_round():
        with gpu.Handle(N, dh, 3, flags=gpu.MG_FLAG_RESERVE_EF) as h:
            h.set_rho(rho)
            h.set_phi(np.zeros(N))
            h.vcycles(2)
            h.compute_ef(download=False)
        with gpu.Handle(N, dh, 3, devices=[0, 0]) as h:
            h.set_rho(rho)
            h.set_phi(np.zeros(N))
            h.vcycles(2)
            h.compute_ef(download=False)
        d = gpu.DistRank(N, dh, 3, 0, 1, device=0)
        try:
            d.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
            d.set_phi(lambda lo, hi: np.zeros((hi - lo,) + N[1:]))
            d.vcycles(1)
            d.compute_ef_owned()
        finally:
            d.close()

    one_round()  # first-use allocations of the runtime
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info(0)[0]
    for _ in range(10):
        one_round()
    torch.cuda.synchronize()
    assert before - torch.cuda.mem_get_info(0)[0] <= (8 << 20)


@pytest.mark.parametrize("n,depth", [((3, 3, 3), 0), ((5, 4, 3), 0), ((6, 6, 6), 1), ((7, 9, 6), 1), ((12, 12, 12), 2),
                                      ((13, 12, 14), 2)], ids=str)
def test_smallest_grids(gpu, orc, n, depth):
    """Three nodes per axis on the coarsest level is the least the loops allow (one interior node):
    the Gauss-Seidel solver (depth 0) and the V-cycle still equal the oracle bit for bit."""
    from util import assert_bit_equal
    dh = orc.main_extents_dh(n)
    g = np.random.default_rng(5)
    rho, phi0 = g.uniform(-1, 1, n) * 8.85e-9, g.uniform(-1, 1, n)
    want = phi0.copy()
    with gpu.Handle(n, dh, depth) as h:
        h.set_rho(rho)
        h.set_phi(phi0)
        if depth == 0:
            got = h.gs_solve(60, 1e-30)
            ref = orc.gs_solve(want, rho, dh, 60, 1e-30, rb=True)
            assert got[:2] == tuple(ref[:2]) and got[2] == pytest.approx(ref[2], rel=1e-12)
        else:
            l2 = h.vcycles(3)
            ref = orc.Solver(n, dh, depth).solve(want, rho, 3, 0.0, rb=True)
            assert l2 == pytest.approx(ref[2], rel=1e-12)
        assert_bit_equal(h.get_phi(), want, "phi")


def test_no_iterations_and_not_a_number(gpu, orc):
    """max_it = 0 does nothing and reports failure like the reference's loop that never runs
    (cpp/PotentialSolver.cpp:3756, :4246-4247); a NaN in rho makes every norm NaN, which never passes
    the tolerance test: the solve runs its max_it cycles and returns false, it does not hang."""
    N = (41, 41, 41)
    dh, rho = orc.main_extents_dh(N), orc.synthetic_rho(N)
    with gpu.Handle(N, dh, 2) as h:
        h.set_rho(rho)
        h.set_phi(np.zeros(N))
        assert h.solve(0, 1e-6) == (False, 0, 0.0)
        assert not h.get_phi().any()
        rho[20, 20, 20] = np.nan
        h.set_rho(rho)
        ok, cycles, l2 = h.solve(5, 1e-6)
        assert (ok, cycles) == (False, 5) and np.isnan(l2)


def test_handles_on_different_host_threads_do_not_disturb_each_other(gpu, orc):
    """One handle per host thread is the contract: four threads, each creating, solving with and
    destroying handles of its own at the same time, get the bits the same solves give one after
    the other."""
    import threading
    from util import assert_bit_equal
    cases = [((129, 129, 129), 3), ((65, 65, 65), 3), ((100, 90, 80), 2), ((257, 257, 257), 5)]

    def run(n, depth):
        with gpu.Handle(n, orc.main_extents_dh(n), depth) as h:
            h.set_rho(orc.synthetic_rho(n))
            h.set_phi(np.zeros(n))
            return h.solve(200, 1e-6), h.get_phi(), h.compute_ef()

    alone = {n: run(n, depth) for n, depth in cases}
    together, errors = {}, []

    def work(n, depth):
        try:
            for rep in range(2):
                together[(n, rep)] = run(n, depth)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=c) for c in cases]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (n, rep), (res, phi, ef) in together.items():
        assert res == alone[n][0]
        assert_bit_equal(phi, alone[n][1], f"phi of {n}, repetition {rep}")
        assert_bit_equal(ef, alone[n][2], f"E of {n}, repetition {rep}")
