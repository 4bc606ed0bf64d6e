"""The slab-partitioned solve driven inside libmg_hip (mg_dist_*, and mg_create with n_gpus > 1):
several ranks -- host threads here -- share the one GPU of the test box and move halo planes by
device-to-device copies (the in-process transport), either as soon as an exchange is started or
held back until its planes are next needed, the two extremes of when RCCL may deliver them on a
multi-GPU node.  The kernels, the slab geometry and the exchange schedule are those of the RCCL
path; the assembled potential must equal the undivided red-black oracle bit for bit."""
import threading

import numpy as np
import pytest

from util import assert_bit_equal

pytestmark = pytest.mark.gpu


def run_ranks_with(gpu, world, n, depth, cut, body, hold_back, nu):
    return run_ranks(gpu, world, n, depth, cut, body, hold_back, nu=nu)


def run_ranks(gpu, world, n, depth, cut, body, hold_back=False, overlap=True, join_timeout=240, deep_halos=True, nu=None):
    """body(rank_object, rank) on `world` threads; returns their results by rank."""
    from multigrid_amd import problem
    N = (n, n, n) if isinstance(n, int) else tuple(n)
    dh = problem.main_extents_dh(N)
    group = gpu.LocalGroup(world, hold_back) if world > 1 else None
    out, err = [None] * world, [None] * world

    def work(r):
        d = None
        try:
            d = gpu.DistRank(N, dh, depth, r, world, device=0, local_group=group, coarse_cut=cut, overlap=overlap,
                             deep_halos=deep_halos, nu=nu)
            d.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
            d.set_phi(lambda lo, hi: np.zeros((hi - lo,) + N[1:]))
            out[r] = body(d, r)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            err[r] = e
        finally:
            if d is not None:
                d.close()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=join_timeout)
    assert not any(t.is_alive() for t in threads), "a rank is stuck"
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("world,n,depth,cut,cycles", [(1, 65, 3, 2, 3), (2, 65, 3, 1, 3), (2, 129, 3, 2, 3),
                                                      (2, 257, 5, 2, 3), (3, 257, 5, 3, 2), (4, 257, 5, 2, 2),
                                                      (2, 513, 5, 3, 2)])
@pytest.mark.parametrize("hold_back", [False, True], ids=["exchange_early", "exchange_late"])
def test_native_partitioned_cycle_equals_undivided_oracle(gpu, orc, world, n, depth, cut, cycles, hold_back):
    if hold_back and (world == 1 or n >= 513):
        pytest.skip("covered by the early-exchange case")

    def body(d, r):
        hist = [d.vcycles(1) for _ in range(cycles)]
        lo, phi = d.get_phi_owned()
        return lo, phi, hist

    res = run_ranks(gpu, world, n, depth, cut, body, hold_back)
    N = (n, n, n)
    dh = orc.main_extents_dh(N)
    want = np.zeros(N)
    _, _, _, hist = orc.Solver(N, dh, depth).solve(want, orc.synthetic_rho(N), cycles, 0.0, rb=True)
    got = np.full(N, np.nan)
    for lo, part, _ in res:
        got[lo:lo + part.shape[0]] = part
    assert_bit_equal(got, want, "phi assembled from the slabs")
    for _, _, h in res:  # every rank reports the all-reduced norm
        assert h == pytest.approx(list(hist), rel=1e-12)
        assert h == res[0][2]


def test_native_solve_stops_on_the_same_iteration_on_every_rank(gpu, orc):
    n, depth, world = 129, 3, 3
    res = run_ranks(gpu, world, n, depth, 2, lambda d, r: d.solve(500, 1e-6) + d.get_phi_owned())
    N = (n, n, n)
    want = np.zeros(N)
    ok, cyc, l2, _ = orc.Solver(N, orc.main_extents_dh(N), depth).solve(want, orc.synthetic_rho(N), 500, 1e-6, rb=True)
    got = np.zeros(N)
    for conv, cycles, norm, lo, part in res:
        assert (conv, cycles) == (ok, cyc)
        assert norm == pytest.approx(l2, rel=1e-12)
        got[lo:lo + part.shape[0]] = part
    assert_bit_equal(got, want, "converged phi assembled from the slabs")


def test_no_overlap_schedule_gives_the_same_bits(gpu, orc):
    n, depth = 257, 5
    res = run_ranks(gpu, 2, n, depth, 2, lambda d, r: (d.vcycles(2),) + d.get_phi_owned(), overlap=False)
    N = (n, n, n)
    want = np.zeros(N)
    orc.Solver(N, orc.main_extents_dh(N), depth).solve(want, orc.synthetic_rho(N), 2, 0.0, rb=True)
    got = np.zeros(N)
    for _, lo, part in res:
        got[lo:lo + part.shape[0]] = part
    assert_bit_equal(got, want, "phi assembled from the slabs")


@pytest.mark.parametrize("devices,n,depth", [([0, 0], 129, 3), ([0, 0, 0], 257, 5)])
def test_multi_gpu_handle_equals_single_gpu_handle(gpu, orc, devices, n, depth):
    """mg_create with n_gpus > 1 (here every slab on GPU 0) behind the ordinary handle calls."""
    from multigrid_amd import problem
    N = (n, n, n)
    dh = problem.main_extents_dh(N)
    rho = problem.synthetic_rho(N)
    with gpu.Handle(N, dh, depth) as one, gpu.Handle(N, dh, depth, devices=devices) as many:
        for h in (one, many):
            h.set_rho(rho)
            h.set_phi(np.zeros(N))
        assert many.vcycles(2) == pytest.approx(one.vcycles(2), rel=1e-12)
        assert_bit_equal(many.get_phi(), one.get_phi(), "two cycles")
        a, b = one.solve(500, 1e-6), many.solve(500, 1e-6)
        assert a[:2] == b[:2] and b[2] == pytest.approx(a[2], rel=1e-12)
        assert_bit_equal(many.get_phi(), one.get_phi(), "converged")
        assert many.last_call_ms() > 0
        with pytest.raises(gpu.MgError):  # no field computed on the current potential yet
            many.get_ef()
        assert_bit_equal(many.compute_ef(), one.compute_ef(), "E from the slabs")
        assert_bit_equal(many.compute_ef(), orc.compute_ef(one.get_phi(), dh), "E against the oracle")
        many.vcycles(1)
        with pytest.raises(gpu.MgError):  # stale after a cycle
            many.get_ef()


def test_rank_failure_does_not_hang_the_group(gpu):
    """A partition the grid cannot honour is reported by the handle call, with the rank named."""
    from multigrid_amd import problem
    N = (65, 65, 65)
    with pytest.raises(gpu.MgError) as e:
        gpu.Handle(N, problem.main_extents_dh(N), 3, devices=[0] * 16, coarse_cut=3)  # 4 planes per rank < 2^3
    assert "rank" in str(e.value) and "planes" in str(e.value)


def test_rccl_transport_in_a_world_of_one(gpu, orc):
    """The RCCL transport on the one GPU of the test box: a communicator of one rank, whose creation
    sends a block to itself, receives it and all-reduces a value on the communication stream (the
    calls the halo exchange, the norm and the gather make), then a solve through that transport."""
    from multigrid_amd import problem
    n, depth = 65, 3
    N = (n, n, n)
    dh = problem.main_extents_dh(N)
    d = gpu.DistRank(N, dh, depth, 0, 1, device=0, unique_id=gpu.dist_unique_id(), coarse_cut=2)
    try:
        d.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
        d.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
        ok, cycles, l2 = d.solve(500, 1e-6)
        lo, phi = d.get_phi_owned()
    finally:
        d.close()
    want = np.zeros(N)
    w_ok, w_cyc, w_l2, _ = orc.Solver(N, dh, depth).solve(want, orc.synthetic_rho(N), 500, 1e-6, rb=True)
    assert (ok, cycles) == (w_ok, w_cyc) and lo == 0
    assert l2 == pytest.approx(w_l2, rel=1e-12)
    assert_bit_equal(phi, want, "phi through the RCCL transport")


def test_an_aborted_rccl_rank_fails_fast_and_can_be_destroyed(gpu):
    """mg_dist_abort on an RCCL rank (ncclCommAbort through the same binding): the next collective call
    returns MG_ERR_COMM instead of queueing transfers, and the rank can still be destroyed."""
    from multigrid_amd import problem
    n, depth = 65, 3
    N = (n, n, n)
    d = gpu.DistRank(N, problem.main_extents_dh(N), depth, 0, 1, device=0, unique_id=gpu.dist_unique_id(), coarse_cut=2)
    try:
        d.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
        d.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
        assert np.isfinite(d.vcycles(1))
        d.abort()
        d.abort()  # idempotent
        with pytest.raises(gpu.MgError) as e:
            d.vcycles(1)
        assert e.value.code == -6 and "given up" in str(e.value)
    finally:
        d.close()


def test_aborting_one_rank_releases_the_ranks_waiting_for_it(gpu):
    """Three in-process ranks; rank 1 never enters the cycle and is aborted instead (what a supervisor
    does with a rank that failed): the others, waiting for its halo planes, return MG_ERR_COMM."""
    def body(d, r):
        if r == 1:
            import time
            time.sleep(0.5)  # the others are inside the exchange by now
            d.abort()
            return "aborted"
        try:
            d.vcycles(2)
        except gpu.MgError as e:
            return e.code
        return "finished"

    assert run_ranks(gpu, 3, 129, 3, 2, body, join_timeout=60) == [-6, "aborted", -6]


def test_a_rank_alone_runs_its_share_of_a_larger_world(gpu):
    """MG_TRANSPORT_NONE: rank 1 and rank 0 of a world of four, each without peers, run the launches
    of their share (every transfer skipped) and return; the plan is that of the full world.  A
    world of one through the same transport moves nothing either way, so it still equals the
    single-GPU handle."""
    from multigrid_amd import problem
    n, depth = 129, 3
    N = (n, n, n)
    dh = problem.main_extents_dh(N)
    from multigrid_amd.dist import SlabPlan
    plan = SlabPlan(N, depth, 4, cut=2)  # the Python twin's arithmetic: both drivers cut the grid alike
    assert plan.bounds[1] == plan.bounds[2] - plan.bounds[1]  # equal shares: every rank runs the whole levels too
    for rank in (1, 0):
        d = gpu.DistRank(N, dh, depth, rank, 4, device=0, coarse_cut=2, alone=True)
        try:
            assert d.own == plan.own(0, "phi", rank) and d.cut == 2
            d.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
            d.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
            assert np.isfinite(d.vcycles(2))
            assert d.stats()["bytes_sent"] == 0 and d.stats()["last_ms"] > 0
        finally:
            d.close()
    d = gpu.DistRank(N, dh, depth, 0, 1, device=0, alone=True)
    try:
        d.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
        d.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
        l2 = d.vcycles(2)
        _, phi = d.get_phi_owned()
    finally:
        d.close()
    with gpu.Handle(N, dh, depth) as h:
        h.set_rho(problem.synthetic_rho(N))
        h.set_phi(np.zeros(N))
        assert h.vcycles(2) == pytest.approx(l2, rel=1e-12)
        assert_bit_equal(phi, h.get_phi(), "phi of a world of one without transport")


@pytest.mark.parametrize("n,depth,world,cut", [((17, 17, 17), 1, 8, 0), ((33, 33, 33), 2, 8, 2), ((21, 9, 11), 1, 5, 0),
                                               ((65, 20, 24), 2, 4, 1), ((65, 65, 65), 3, 16, 1), ((129, 40, 40), 3, 3, 3)],
                         ids=str)
def test_extreme_partitions_equal_the_undivided_handle(gpu, n, depth, world, cut):
    """Slabs of a single unit of 2^cut planes, sixteen ranks, non-cubic grids, every level
    distributed: potential and field of the multi-GPU handle (all slabs on GPU 0) equal the
    single-GPU handle's bit for bit."""
    from multigrid_amd import problem
    dh = problem.main_extents_dh(n)
    g = np.random.default_rng(7)
    rho, phi0 = g.uniform(-1, 1, n) * 8.85e-9, g.uniform(-1, 1, n)
    with gpu.Handle(n, dh, depth) as h:
        h.set_rho(rho)
        h.set_phi(phi0)
        l2 = h.vcycles(3)
        phi, ef = h.get_phi(), h.compute_ef()
    with gpu.Handle(n, dh, depth, devices=[0] * world, coarse_cut=cut) as h:
        h.set_rho(rho)
        h.set_phi(phi0)
        assert h.vcycles(3) == pytest.approx(l2, rel=1e-12)
        assert_bit_equal(h.get_phi(), phi, "phi of the slabs")
        assert_bit_equal(h.compute_ef(), ef, "E of the slabs")


@pytest.mark.parametrize("n,depth,world,cut,nu", [
    ((129, 129, 129), 3, 8, 1, (1, 2, 1, 6)),        # levels 1 and 2 cut down (65 and 33 planes), tiles
    ((257, 257, 257), 4, 8, 1, (1, 3, 2, 1, 9)),     # level 1 = 129^3 cut down: tiles on ranges of planes the march could take
    ((257, 257, 257), 4, 8, 2, None),                # the default schedule: 65 planes of level 2 cut down to 31
    ((257, 140, 150), 3, 16, 2, (1, 1, 2, 5)),       # sixteen ranks, non-cubic
    ((513, 72, 136), 4, 8, 2, (1, 5, 3, 2, 7)),      # 129 planes of 19 x 35 cut down
    ((257, 600, 560), 3, 8, 1, (1, 3, 2, 6)),        # 129 planes of 301 x 281: the marching kernel on ranges of a whole array
], ids=str)
def test_whole_levels_cut_down_to_what_a_rank_needs(gpu, n, depth, world, cut, nu):
    """Every rank runs the levels below the partition, each only on the planes from which its own
    slab can be reached: two planes more per side for every sweep still to come, the level below
    on what the prolongation onto that range reads.  Potential after three cycles, then after a
    solve, against the undivided handle, bit for bit."""
    from multigrid_amd import problem
    dh = problem.main_extents_dh(n)
    g = np.random.default_rng(19)
    rho, phi0 = g.uniform(-1, 1, n) * 8.85e-9, g.uniform(-1, 1, n)
    out = []
    for devices in (None, [0] * world):
        with gpu.Handle(n, dh, depth, nu=nu, devices=devices, coarse_cut=cut if devices else 0) as h:
            h.set_rho(rho)
            h.set_phi(phi0)
            l2 = h.vcycles(3)
            phi3 = h.get_phi()
            ok, cycles, _ = h.solve(40, 1e-3)
            out.append((l2, phi3, ok, cycles, h.get_phi()))
    assert out[1][0] == pytest.approx(out[0][0], rel=1e-12)
    assert_bit_equal(out[1][1], out[0][1], "phi of the slabs after three cycles")
    assert out[1][2:4] == out[0][2:4]
    assert_bit_equal(out[1][4], out[0][4], "phi of the slabs after the solve")


@pytest.mark.parametrize("devices,cut", [([0, 0], 0), ([0, 0, 0], 2), ([0] * 4, 1)], ids=str)
def test_sequence_of_warm_started_solves_on_slabs(gpu, devices, cut):
    """The reference's time loop calls solve() again and again on a changing rho with phi and the
    coarse corrections left where they were (cpp/Main.cpp:151-166): a solve, a new rho, a solve, an
    edited phi, a solve -- the slabs take the undivided handle's cycle counts and end on its bits."""
    from multigrid_amd import problem
    n, depth = (129, 129, 129), 3
    dh = problem.main_extents_dh(n)
    rho1 = problem.synthetic_rho(n)
    rho2 = rho1 * (1 + 0.05 * np.random.default_rng(3).uniform(-1, 1, n))

    def sequence(h):
        out = []
        h.set_rho(rho1)
        h.set_phi(np.zeros(n))
        out.append(h.solve(300, 1e-6)[:2])
        h.set_rho(rho2)
        out.append(h.solve(300, 1e-6)[:2])
        phi = h.get_phi()
        out.append(phi.copy())
        phi[1:-1, 1:-1, 1:-1] *= 0.5
        h.set_phi(phi)
        out.append(h.solve(300, 1e-6)[:2])
        return out + [h.get_phi(), h.compute_ef()]

    with gpu.Handle(n, dh, depth) as h:
        want = sequence(h)
    with gpu.Handle(n, dh, depth, devices=devices, coarse_cut=cut) as h:
        got = sequence(h)
    assert want[1][1] < want[0][1]  # the warm start pays
    for a, b in zip(got, want):
        if isinstance(b, tuple):
            assert a == b
        else:
            assert_bit_equal(a, b, "field after the sequence")


def _partition_cases():
    import os
    g = np.random.default_rng(int(os.environ.get("MG_FUZZ_SEED", "20261111")) + 2)
    out = []
    for _ in range(6 * int(os.environ.get("MG_FUZZ_MULT", "1"))):
        n = tuple(int(x) for x in g.integers(20, int(g.choice([60, 140, 270])), size=3))
        deepest = 0
        while (min(n) >> (deepest + 1)) >= 3:
            deepest += 1
        depth = int(g.integers(1, min(deepest, 5) + 1))
        out.append((n, depth, int(g.integers(2, 9)), int(g.integers(0, depth + 1)), bool(g.integers(0, 2))))
    return out


@pytest.mark.parametrize("n,depth,world,cut,solve", _partition_cases(), ids=str)
def test_random_partitions_equal_the_undivided_handle(gpu, n, depth, world, cut, solve):
    """Random non-cubic grids cut into a random number of slabs at a random level (cut 0 = the
    library's choice): cycles or a solve on the slabs end on the undivided handle's bits.  A grid that
    cannot be cut that way must be refused by mg_create, not mis-solved."""
    from multigrid_amd import problem
    dh = problem.main_extents_dh(n)
    g = np.random.default_rng(17 + sum(n))
    rho, phi0 = g.uniform(-1, 1, n) * 8.85e-9, g.uniform(-1, 1, n)

    def run(h):
        h.set_rho(rho)
        h.set_phi(phi0)
        head = h.solve(40, 1e-3)[:2] if solve else None
        if not solve:
            h.vcycles(2)
        return head, h.get_phi(), h.compute_ef()

    try:
        h = gpu.Handle(n, dh, depth, devices=[0] * world, coarse_cut=cut)
    except gpu.MgError as e:
        assert e.code == -1 and "planes over" in str(e)  # too few planes for that many slabs of 2^cut
        return
    with h:
        got = run(h)
    with gpu.Handle(n, dh, depth) as h:
        want = run(h)
    assert got[0] == want[0]
    assert_bit_equal(got[1], want[1], "phi of the slabs")
    assert_bit_equal(got[2], want[2], "E of the slabs")


def _cut_down_cases():
    import os
    g = np.random.default_rng(int(os.environ.get("MG_FUZZ_SEED", "20261111")) + 4)
    out = []
    for _ in range(5 * int(os.environ.get("MG_FUZZ_MULT", "1"))):
        # a long first axis over many ranks and short sweep schedules: the ranges a rank needs of the
        # whole levels are then well below half a level
        n = (int(g.integers(120, 400)), int(g.integers(12, 150)), int(g.integers(12, 150)))
        deepest = 0
        while (min(n) >> (deepest + 1)) >= 3:
            deepest += 1
        depth = int(g.integers(2, min(deepest, 5) + 1)) if deepest >= 2 else deepest
        cut = int(g.integers(1, depth + 1))
        nu = tuple(int(x) for x in g.integers(0, 4, size=depth)) + (int(g.integers(1, 12)),)
        out.append((n, depth, int(g.integers(4, 17)), cut, nu))
    return out


@pytest.mark.parametrize("n,depth,world,cut,nu", _cut_down_cases(), ids=str)
def test_random_sweep_schedules_on_cut_down_levels(gpu, n, depth, world, cut, nu):
    """Random grids with a long first axis, many ranks, random short sweep schedules (zero sweeps on
    a level included) and cuts: whatever planes of the whole levels each rank decides to compute, the
    potential equals the undivided handle's."""
    from multigrid_amd import problem
    if depth < 1:
        pytest.skip("grid too small for a coarse level")
    dh = problem.main_extents_dh(n)
    g = np.random.default_rng(23 + sum(n))
    rho, phi0 = g.uniform(-1, 1, n) * 8.85e-9, g.uniform(-1, 1, n)
    try:
        h = gpu.Handle(n, dh, depth, nu=nu, devices=[0] * world, coarse_cut=cut)
    except gpu.MgError as e:
        assert e.code == -1 and "planes over" in str(e)
        return
    with h:
        h.set_rho(rho)
        h.set_phi(phi0)
        l2 = h.vcycles(3)
        got = h.get_phi()
    with gpu.Handle(n, dh, depth, nu=nu) as h:
        h.set_rho(rho)
        h.set_phi(phi0)
        assert h.vcycles(3) == pytest.approx(l2, rel=1e-12)
        assert_bit_equal(got, h.get_phi(), "phi of the slabs")


def _undivided(gpu, n, depth, cycles, nu=None, rho=None, phi0=None):
    from multigrid_amd import problem
    with gpu.Handle(n, problem.main_extents_dh(n), depth, nu=nu) as h:
        h.set_rho(problem.synthetic_rho(n) if rho is None else rho)
        h.set_phi(np.zeros(n) if phi0 is None else phi0)
        l2 = h.vcycles(cycles)
        return l2, h.get_phi()


@pytest.mark.parametrize("world,n,depth,cut,fits", [(2, 257, 5, 3, True), (3, 257, 5, 3, True), (4, 257, 5, 3, False),
                                                    (4, 513, 5, 3, True), (2, 129, 4, 2, True), (2, 513, 5, 4, True)])
@pytest.mark.parametrize("hold_back", [False, True], ids=["exchange_early", "exchange_late"])
@pytest.mark.parametrize("deep_halos", [True, False], ids=["deep_halos", "exchange_every_sweep"])
def test_deep_halos_on_coarse_distributed_levels(gpu, world, n, depth, cut, fits, hold_back, deep_halos):
    """Levels 1..cut-1 with ONE exchange of R_l per cycle: eps_l is formed on the slab widened by two
    planes per sweep of the level and swept on ranges that shrink from sweep to sweep (in pairs where
    the kernels take two sweeps per launch); where a neighbour holds too few planes for that
    (4 ranks at 257^3: 16 planes of level 2, 17 needed) the rank exchanges after every sweep instead.
    Either way, with every exchange completed early or as late as the schedule allows, three cycles
    end on the undivided handle's bits."""
    if n >= 513 and (hold_back or not deep_halos):
        pytest.skip("covered at 257^3")
    N = (n, n, n)

    def body(d, r):
        assert d.deep_halos == (fits and deep_halos)
        return (d.vcycles(3),) + d.get_phi_owned()

    res = run_ranks(gpu, world, n, depth, cut, body, hold_back, deep_halos=deep_halos)
    l2, want = _undivided(gpu, N, depth, 3)
    got = np.full(N, np.nan)
    for norm, lo, part in res:
        assert norm == pytest.approx(l2, rel=1e-12)
        got[lo:lo + part.shape[0]] = part
    assert_bit_equal(got, want, "phi assembled from the slabs")


@pytest.mark.parametrize("n,depth,world,cut,nu", [((2049, 24, 28), 3, 16, 3, None),      # sixteen ranks, 32 planes of level 2 each
                                                  ((1025, 40, 300), 3, 5, 3, (2, 4, 3, 7)),  # unequal shares, the fine level swept twice
                                                  ((513, 300, 56), 4, 3, 2, (1, 6, 1, 1, 9))], ids=str)
def test_deep_halos_with_many_ranks_and_unequal_shares(gpu, n, depth, world, cut, nu):
    """Deep halos where every rank but the outer two has two neighbours to fetch from and to serve,
    where the shares differ by a unit of 2^cut planes, and where the fine level needs several sweeps
    (its exchanges after every sweep remain): every exchange held back, three cycles, the undivided
    handle's bits."""
    def body(d, r):
        assert d.deep_halos
        return (d.vcycles(3),) + d.get_phi_owned()

    res = run_ranks(gpu, world, n, depth, cut, body, hold_back=True, nu=nu)
    l2, want = _undivided(gpu, n, depth, 3, nu=nu)
    got = np.full(n, np.nan)
    for norm, lo, part in res:
        assert norm == pytest.approx(l2, rel=1e-12)
        got[lo:lo + part.shape[0]] = part
    assert_bit_equal(got, want, "phi assembled from the slabs")


def _deep_cases():
    import os
    g = np.random.default_rng(int(os.environ.get("MG_FUZZ_SEED", "20261111")) + 5)
    out = []
    for _ in range(6 * int(os.environ.get("MG_FUZZ_MULT", "1"))):
        # a long first axis, few ranks, short schedules: the planes the sweeps reach lie on the neighbours
        wide = bool(g.integers(0, 2))  # rows of 256 and more take the marching kernels and the on-load prolongation
        n = (int(g.integers(130, 420)), int(g.integers(260, 300) if wide else g.integers(24, 200)),
             int(g.integers(260, 300) if wide else g.integers(24, 200)))
        deepest = 0
        while (min(n) >> (deepest + 1)) >= 3:
            deepest += 1
        depth = int(g.integers(2, min(deepest, 4) + 1))
        cut = int(g.integers(2, depth + 1))
        nu = (int(g.integers(1, 3)),) + tuple(int(x) for x in g.integers(0, 5, size=depth - 1)) + (int(g.integers(1, 12)),)
        out.append((n, depth, int(g.integers(2, 5)), cut, nu, bool(g.integers(0, 2))))
    return out


@pytest.mark.parametrize("n,depth,world,cut,nu,hold_back", _deep_cases(), ids=str)
def test_random_schedules_on_deep_halos(gpu, n, depth, world, cut, nu, hold_back):
    """Random grids with a long first axis, cuts at level 2 and below and random sweep schedules (zero,
    odd and even counts): the library decides for deep halos exactly when the plan's arithmetic says
    that every rank finds its planes on its two neighbours, and the potential after three cycles
    equals the undivided handle's."""
    from multigrid_amd.dist import SlabPlan
    try:
        plan = SlabPlan(n, depth, world, cut)
    except ValueError:
        plan = None

    def body(d, r):
        assert d.deep_halos == plan.deep_fits(nu)
        return (d.vcycles(3),) + d.get_phi_owned()

    N = tuple(n)
    try:
        res = run_ranks_with(gpu, world, N, depth, cut, body, hold_back, nu)
    except gpu.MgError as e:
        assert plan is None and e.code == -1 and "planes over" in str(e)
        return
    assert plan is not None
    l2, want = _undivided(gpu, N, depth, 3, nu=nu)
    got = np.full(N, np.nan)
    for norm, lo, part in res:
        assert norm == pytest.approx(l2, rel=1e-12)
        got[lo:lo + part.shape[0]] = part
    assert_bit_equal(got, want, "phi assembled from the slabs")


def _late_exchange_cases():
    import os
    g = np.random.default_rng(int(os.environ.get("MG_FUZZ_SEED", "20261111")) + 3)
    out = []
    for _ in range(4 * int(os.environ.get("MG_FUZZ_MULT", "1"))):
        n = (int(g.integers(40, 200)), int(g.integers(130, 280)), int(g.integers(130, 280)))
        depth = int(g.integers(1, 4))
        out.append((n, depth, int(g.integers(2, 5)), int(g.integers(1, depth + 1))))
    return out


@pytest.mark.parametrize("n,depth,world,cut", _late_exchange_cases(), ids=str)
def test_overlapped_schedule_with_every_exchange_held_back_on_random_grids(gpu, n, depth, world, cut):
    """Grids wide enough for the out-of-place sweep (the edge / exchange / interior schedule and the
    split residual pass) cut at random, every halo exchange completed only when its planes are next
    read: a wait missing from the schedule would show as wrong bits against the undivided handle."""
    from multigrid_amd import problem
    try:
        res = run_ranks(gpu, world, n, depth, cut, lambda d, r: (d.vcycles(2),) + d.get_phi_owned(), hold_back=True)
    except gpu.MgError as e:
        assert e.code == -1 and "planes over" in str(e)
        return
    got = np.full(n, np.nan)
    for _, lo, part in res:
        got[lo:lo + part.shape[0]] = part
    with gpu.Handle(n, problem.main_extents_dh(n), depth) as h:
        h.set_rho(problem.synthetic_rho(n))
        h.set_phi(np.zeros(n))
        l2 = h.vcycles(2)
        assert_bit_equal(got, h.get_phi(), "phi assembled from the slabs")
    assert all(r[0] == pytest.approx(l2, rel=1e-12) for r in res)


def _calls(log):
    """A rank's record split into its calls: [(what, array, (lo, hi), [(send?, peer, array, lo, hi), ...])]."""
    calls = []
    for what, peer, tag, lo, hi in log:
        if what in (0, 3, 4):
            calls.append((what, tag, (lo, hi), []))
        else:
            assert what in (1, 2) and calls and calls[-1][0] in (0, 4), "a transfer outside a call"
            calls[-1][3].append((what == 1, peer, tag, lo, hi))
    return calls


def _check_records(logs):
    """Every rank made the same calls in the same order, and within each call every send has exactly
    one receive on its peer for the same planes of the same array (an all-gather sends from the
    slab R_l, array 16 l + 3, into the whole-level copy, 16 l + 6)."""
    world = len(logs)
    per_rank = [_calls(l) for l in logs]
    shape = [(what, tag) for what, tag, _, _ in per_rank[0]]
    for r in range(world):
        assert [(what, tag) for what, tag, _, _ in per_rank[r]] == shape, f"rank {r} makes other calls than rank 0"
    widest = 0
    for c, (what, tag) in enumerate(shape):
        sends, recvs = {}, {}
        for r in range(world):
            _, _, (olo, ohi), msgs = per_rank[r][c]
            for send, peer, mtag, lo, hi in msgs:
                assert 0 <= peer < world and peer != r and lo < hi
                book = sends if send else recvs
                key = (r, peer) if send else (peer, r)   # (from, to)
                assert key not in book, f"call {c}: two transfers {key} in one group"
                book[key] = (mtag, lo, hi)
                if what == 0:
                    assert abs(peer - r) == 1 and mtag == tag
                    if send:
                        assert olo <= lo and hi <= ohi, "a rank sends planes it does not own"
                    else:
                        assert hi <= olo or lo >= ohi, "a rank receives planes it owns"
                    widest = max(widest, hi - lo)
                else:
                    assert (lo, hi) == (olo, ohi) if send else True
        assert set(sends) == set(recvs), f"call {c} ({what}, array {tag}): unmatched transfers {set(sends) ^ set(recvs)}"
        for key, (stag, lo, hi) in sends.items():
            rtag, rlo, rhi = recvs[key]
            assert (lo, hi) == (rlo, rhi), f"call {c}: {key} sends planes {lo}..{hi}, the peer expects {rlo}..{rhi}"
            assert rtag == (stag if what == 0 else stag + 3)
        if what == 4:
            assert len(sends) == world * (world - 1)
    return shape, widest


@pytest.mark.parametrize("n,depth,world,cut,nu,overlap,deep", [
    (1025, 6, 8, 0, None, True, True),      # the scaling run's plans
    (1025, 6, 4, 0, None, True, True),
    (1025, 6, 2, 0, None, True, True),
    (513, 5, 8, 0, None, True, True),
    (257, 5, 3, 3, None, True, True),       # unequal shares, two partitioned coarse levels
    (257, 5, 3, 3, None, False, True),      # exchanges after whole sweeps
    (257, 5, 3, 3, None, True, False),      # an exchange after every sweep on every level
    (257, 4, 2, 3, [2, 3, 4, 5, 50], True, True),
    ((129, 40, 40), 3, 5, 2, None, True, True),
], ids=str)
def test_every_send_of_every_rank_has_its_receive(gpu, n, depth, world, cut, nu, overlap, deep):
    """The RCCL transport cannot run between the ranks of a world on the one GPU of the test box, and
    the in-process transport that stands in for it is driven by the receiver alone.  Here every
    rank of the world runs by itself (MG_TRANSPORT_NONE) -- two cycles, a solve and computeEF -- and
    records the calls it would have made over RCCL, built from the message lists the RCCL transport
    issues; the records are then matched against each other: same calls in the same order on every
    rank (a collective or a grouped exchange that one rank skips is a hang), and per call every send
    met by one receive for the same global planes of the same array (a count that differs is a hang
    or silent corruption)."""
    from multigrid_amd import problem
    N = (n, n, n) if isinstance(n, int) else tuple(n)
    dh = problem.main_extents_dh(N)
    logs, deep_seen = [], None
    for r in range(world):
        d = gpu.DistRank(N, dh, depth, r, world, device=0, coarse_cut=cut, nu=nu, overlap=overlap, deep_halos=deep, alone=True)
        try:
            zeros = lambda lo, hi: np.zeros((hi - lo,) + N[1:])  # noqa: E731 -- the numbers mean nothing here
            d.set_rho(zeros)
            d.set_phi(zeros)
            d.vcycles(2)
            d.solve(2, 0.0)
            gpu._check(gpu.load().mg_dist_compute_ef(d._d))   # (E stays on the device)
            logs.append(d.comm_log())
            assert deep_seen in (None, d.deep_halos)
            deep_seen = d.deep_halos
        finally:
            d.close()
    shape, widest = _check_records(logs)
    kinds = [what for what, _ in shape]
    assert kinds.count(3) == 4                # one norm per cycle: 2 + 2
    assert kinds.count(4) == 4                # one all-gather of R_cut per cycle
    assert (0, 0) in shape or (0, 1) in shape  # the fine grid's halo planes
    if deep_seen:
        # one exchange of R_l per cycle and partitioned coarse level, more than one sweep's planes wide
        cut_used = [tag // 16 for what, tag in shape if what == 4][0]
        assert cut_used >= 2 and widest > 2
        assert not any(what == 0 and tag % 16 in (4, 5) for what, tag in shape), "eps_l is never exchanged on deep halos"
    elif cut and cut > 1:
        assert any(what == 0 and tag % 16 in (4, 5) for what, tag in shape)


def test_records_match_on_random_plans(gpu):
    """The same matching of every rank's record on random grids, depths, cuts, world sizes up to
    eight, sweep schedules with zeros, and both exchange policies: small grids, so that many plans
    (also those the library must refuse) are seen."""
    import os
    from multigrid_amd import problem
    g = np.random.default_rng(int(os.environ.get("MG_FUZZ_SEED", "20261111")) + 9)
    ran = deep = refused = 0
    for _ in range(40 * int(os.environ.get("MG_FUZZ_MULT", "1"))):
        N = (int(g.integers(17, 260)), int(g.integers(9, 70)), int(g.integers(9, 70)))
        deepest = 0
        while (min(N) >> (deepest + 1)) >= 3:
            deepest += 1
        depth = int(g.integers(1, min(deepest, 5) + 1))
        cut = int(g.integers(0, depth + 1))
        world = int(g.integers(2, 9))
        nu = [int(g.integers(1, 3))] + [int(x) for x in g.integers(0, 5, size=depth - 1)] + [int(g.integers(1, 8))]
        overlap, deep_halos = bool(g.integers(0, 2)), bool(g.integers(0, 4))
        dh = problem.main_extents_dh(N)
        logs, flags = [], set()
        try:
            for r in range(world):
                d = gpu.DistRank(N, dh, depth, r, world, device=0, coarse_cut=cut, nu=nu, overlap=overlap,
                                 deep_halos=deep_halos, alone=True)
                try:
                    zeros = lambda lo, hi: np.zeros((hi - lo,) + N[1:])  # noqa: E731
                    d.set_rho(zeros)
                    d.set_phi(zeros)
                    d.vcycles(2)
                    gpu._check(gpu.load().mg_dist_compute_ef(d._d))
                    logs.append(d.comm_log())
                    flags.add(d.deep_halos)
                finally:
                    d.close()
        except gpu.MgError as e:
            assert e.code == -1 and "planes over" in str(e) and not logs, (N, depth, world, cut, str(e))
            refused += 1
            continue
        assert len(flags) == 1, "the ranks of one world disagree about deep halos"
        try:
            _check_records(logs)
        except AssertionError as e:
            raise AssertionError(f"{N} depth {depth} world {world} cut {cut} nu {nu} overlap {overlap} deep {deep_halos}: {e}")
        ran += 1
        deep += flags.pop()
    print(f"plans matched: {ran} ({deep} on deep halos), refused: {refused}")
    assert ran >= 15 and refused >= 1 and deep >= 3, (ran, refused, deep)
