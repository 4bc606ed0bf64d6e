"""The slab-partitioned solve on CPU: the product's orchestration (multigrid_amd/dist.py: plan,
halo exchange schedule, coarse gather/scatter, all-reduce) runs over gloo with world_size 2, 3, 4 and 8,
with the oracle's slab-form operators standing in for the GPU kernels, and must reproduce the
undivided red-black oracle bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import assert_bit_equal


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, depth, cut, cycles, out_dir, oop=False, nu=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multigrid_amd import dist as mgd, problem
        from oracle_ops import OracleOps
        N = (n, n, n)
        s = mgd.DistSolver(N, problem.main_extents_dh(N), depth, OracleOps(bool(oop)), mgd.Comm(defer=oop == 2), cut=cut, nu=nu)
        s.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
        s.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
        hist = []
        for _ in range(cycles):  # the correction of one iteration rides on the next one's first sweep
            s.smooth_and_residual()
            hist.append(s.l2())
            s.coarse_correction()
        lo, phi = s.get_phi_owned()
        np.save(os.path.join(out_dir, f"phi_{rank}.npy"), phi)
        np.save(os.path.join(out_dir, f"lo_{rank}.npy"), np.array([lo]))
        if rank == 0:
            np.save(os.path.join(out_dir, "hist.npy"), np.array(hist))
        if nu is not None:
            first, _ = s._coarse_ranges()
            np.save(os.path.join(out_dir, f"first_{rank}.npy"), np.array([first[l][1] - first[l][0] for l in range(s.cut, depth + 1)]))
            return
        # the solve loop: all ranks must agree on the stopping iteration
        s2 = mgd.DistSolver(N, problem.main_extents_dh(N), depth, OracleOps(bool(oop)), mgd.Comm(defer=oop == 2), cut=cut)
        s2.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
        s2.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
        ok, cyc, l2 = s2.solve(200, 1e-2)
        np.save(os.path.join(out_dir, f"solve_{rank}.npy"), np.array([float(ok), cyc, l2]))
    finally:
        dist.destroy_process_group()


CASES = [(w, n, d, c, oop) for (w, n, d, c) in [(2, 33, 2, 1), (2, 33, 3, 2), (3, 65, 3, 2), (2, 41, 2, 2)]
         for oop in (0, 1, 2)]
# more ranks (every rank but the first and last has two neighbours; 8 is the node's GPU count)
CASES += [(4, 65, 3, 1, 1), (8, 65, 3, 2, 1), (8, 65, 3, 2, 2)]
OOP_IDS = {0: "in_place", 1: "edges_exchange_interior", 2: "exchange_completes_late"}


@pytest.mark.parametrize("world,n,depth,cut,oop", CASES,
                         ids=[f"{w}-{n}-{d}-{c}-{OOP_IDS[o]}" for (w, n, d, c, o) in CASES])
def test_partitioned_cycle_equals_undivided_oracle(orc, tmp_path, world, n, depth, cut, oop):
    """oop: the operators behave like the GPU's out-of-place sweep kernel, which makes the solver
    issue every sweep (and the fine residual) as edge planes, halo exchange, interior.  With RCCL
    the exchange completes at some point while the interior runs: the blocking transport is the
    earliest case, oop == 2 delays every started exchange until its slab is next needed."""
    cycles = 3
    mp.spawn(_worker, args=(world, free_port(), n, depth, cut, cycles, str(tmp_path), oop), nprocs=world,
             join=True)
    N = (n, n, n)
    dh = orc.main_extents_dh(N)
    rho = orc.synthetic_rho(N)
    want = np.zeros(N)
    ok, cyc, l2, hist = orc.Solver(N, dh, depth).solve(want, rho, cycles, 0.0, rb=True)
    got = np.zeros(N)
    for r in range(world):
        lo = int(np.load(tmp_path / f"lo_{r}.npy")[0])
        part = np.load(tmp_path / f"phi_{r}.npy")
        got[lo:lo + part.shape[0]] = part
    assert_bit_equal(got, want, "phi assembled from the slabs")
    assert np.load(tmp_path / "hist.npy") == pytest.approx(hist, rel=1e-12)
    # solve(): same stopping iteration as the undivided red-black oracle, on every rank
    ref = np.zeros(N)
    ok, cyc, l2, _ = orc.Solver(N, dh, depth).solve(ref, rho, 200, 1e-2, rb=True)
    for r in range(world):
        s = np.load(tmp_path / f"solve_{r}.npy")
        assert (bool(s[0]), int(s[1])) == (ok, cyc)
        assert s[2] == pytest.approx(l2, rel=1e-12)


@pytest.mark.parametrize("world,n,depth,cut,nu,oop", [(8, 129, 3, 1, (1, 2, 1, 6), 1), (8, 129, 3, 1, (1, 2, 1, 6), 0),
                                                      (4, 65, 3, 1, (2, 1, 1, 4), 2), (8, 129, 4, 2, (1, 3, 1, 1, 5), 1)],
                         ids=str)
def test_whole_levels_cut_down_to_what_a_rank_needs(orc, tmp_path, world, n, depth, cut, nu, oop):
    """Every rank runs the levels below the partition, each only on the planes from which its own
    slab can be reached (two planes more per side for every sweep still to come).  Short sweep
    schedules make those ranges small on small grids; the assembled phi must not change."""
    cycles = 3
    mp.spawn(_worker, args=(world, free_port(), n, depth, cut, cycles, str(tmp_path), oop, nu), nprocs=world, join=True)
    N = (n, n, n)
    want = np.zeros(N)
    _, _, _, hist = orc.Solver(N, orc.main_extents_dh(N), depth, nu=nu).solve(want, orc.synthetic_rho(N), cycles, 0.0, rb=True)
    got = np.zeros(N)
    for r in range(world):
        lo = int(np.load(tmp_path / f"lo_{r}.npy")[0])
        part = np.load(tmp_path / f"phi_{r}.npy")
        got[lo:lo + part.shape[0]] = part
    assert_bit_equal(got, want, "phi assembled from the slabs")
    assert np.load(tmp_path / "hist.npy") == pytest.approx(hist, rel=1e-12)
    # the case does exercise cut-down levels: some rank runs the first whole level on less than all of it
    whole = (n >> cut) + 1
    assert min(int(np.load(tmp_path / f"first_{r}.npy")[0]) for r in range(world)) < whole


def _deep_worker(rank, world, port, n, depth, cut, nu, cycles, out_dir, oop, deep):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multigrid_amd import dist as mgd, problem
        from oracle_ops import OracleOps
        N = (n, n, n)
        s = mgd.DistSolver(N, problem.main_extents_dh(N), depth, OracleOps(bool(oop), pairs=oop == 3),
                           mgd.Comm(defer=oop >= 2), cut=cut, nu=nu, deep_halos=deep)
        assert s.deep == deep
        s.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
        s.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
        l2 = s.vcycles(cycles)
        lo, phi = s.get_phi_owned()
        np.save(os.path.join(out_dir, f"phi_{rank}.npy"), phi)
        np.save(os.path.join(out_dir, f"lo_{rank}.npy"), np.array([lo, s.comm.bytes_sent]))
        if rank == 0:
            np.save(os.path.join(out_dir, "l2.npy"), np.array([l2]))
    finally:
        dist.destroy_process_group()


DEEP_CASES = [(2, 65, 3, 3, (1, 1, 1, 4), 0), (2, 65, 3, 3, (1, 1, 1, 4), 2), (2, 65, 3, 2, (2, 3, 2, 4), 1),
              (4, 129, 4, 3, (1, 2, 1, 1, 5), 3), (3, 129, 4, 3, (1, 3, 0, 2, 5), 3), (4, 129, 4, 2, (1, 2, 3, 1, 5), 2)]


@pytest.mark.parametrize("world,n,depth,cut,nu,oop", DEEP_CASES, ids=str)
def test_deep_halos_equal_undivided_oracle(orc, tmp_path, world, n, depth, cut, nu, oop):
    """Levels 1..cut-1 on deep halos: one exchange of R_l per cycle, eps_l formed on the slab widened
    by two planes per sweep, ranges shrinking from sweep to sweep, no exchange between the sweeps.
    oop 0: in-place operators; 1: out-of-place (on-load prolongation); 2: every exchange completed as
    late as the schedule allows; 3: also two sweeps per call where the stored planes allow (as the
    GPU's pair kernel).  The same bits as the undivided oracle, and as the schedule that exchanges
    after every sweep -- with fewer messages."""
    cycles = 3
    runs = {}
    for deep in (True, False):
        out = tmp_path / str(deep)
        out.mkdir()
        mp.spawn(_deep_worker, args=(world, free_port(), n, depth, cut, nu, cycles, str(out), oop, deep), nprocs=world, join=True)
        runs[deep] = out
    N = (n, n, n)
    want = np.zeros(N)
    _, _, _, hist = orc.Solver(N, orc.main_extents_dh(N), depth, nu=nu).solve(want, orc.synthetic_rho(N), cycles, 0.0, rb=True)
    for deep, out in runs.items():
        got = np.zeros(N)
        for r in range(world):
            lo = int(np.load(out / f"lo_{r}.npy")[0])
            part = np.load(out / f"phi_{r}.npy")
            got[lo:lo + part.shape[0]] = part
        assert_bit_equal(got, want, f"phi assembled from the slabs (deep halos: {deep})")
        assert np.load(out / "l2.npy")[0] == pytest.approx(hist[-1], rel=1e-12)


def test_deep_halo_ranges_nest_and_fit(mg):
    """The arithmetic of the deep halos at the sizes of the scaling runs: every rank's ranges lie on
    itself and its two neighbours, what one rank sends is what its neighbour receives, and a level's
    range covers what the prolongation onto the level above reads."""
    from multigrid_amd.dist import SlabPlan, schedule
    for n0, depth, world, cut, fits in [(1025, 6, 8, None, True), (1025, 6, 4, None, True), (1025, 6, 2, None, True),
                                        (2049, 7, 8, None, True), (513, 5, 8, None, True),
                                        (513, 5, 8, 3, False),     # 16 planes of level 2 per rank, 17 needed of a neighbour
                                        (257, 5, 2, None, False)]:  # no coarse level is distributed
        p = SlabPlan((n0,) * 3, depth, world, cut)
        nu = schedule(depth)
        assert p.deep_fits(nu) == fits, (n0, world)
        if not fits:
            continue
        ranges = [p.deep_ranges(r, nu) for r in range(world)]
        for r in range(world):
            first, last = ranges[r]
            need = p.stored(0, "phi", r, 2)
            for l in range(1, p.cut):
                lo, hi = p.own(l, "eps", r)
                assert first[l][0] <= last[l][0] <= lo and hi <= last[l][1] <= first[l][1]
                # the prolongation onto fine plane f reads coarse plane f/2, and f/2 + 1 when f is odd
                f = need[1] - 1
                assert last[l][0] <= need[0] >> 1 and last[l][1] >= min((f >> 1) + 1 + (f & 1), p.planes(l, "eps"))
                need = first[l]
                if r + 1 < world:  # what this rank's range reaches into the next rank's planes, and back
                    assert first[l][1] <= p.own(l, "eps", r + 1)[1]
                    assert ranges[r + 1][0][l][0] >= lo


def test_plan_partitions_every_level(mg):
    from multigrid_amd.dist import SlabPlan
    for n0, depth, world in [(1025, 6, 8), (1025, 6, 2), (513, 5, 4), (2049, 7, 8), (129, 3, 2)]:
        p = SlabPlan((n0,) * 3, depth, world)
        assert 1 <= p.cut <= depth
        for l in range(p.cut + 1):
            for kind in ("R", "eps"):
                covered = []
                for r in range(world):
                    lo, hi = p.own(l, kind, r)
                    assert lo < hi
                    covered += list(range(lo, hi))
                    if r < world - 1:  # coarse slabs nest: boundaries halve exactly
                        assert (p.bounds[r + 1] >> l) << l == p.bounds[r + 1]
                assert covered == list(range(p.planes(l, kind)))
        # residual+restrict needs even slab starts on the fine level
        assert all(b % 2 == 0 for b in p.bounds[:-1])


def test_plan_rejects_impossible_partitions(mg):
    from multigrid_amd.dist import SlabPlan
    with pytest.raises(ValueError):
        SlabPlan((33, 33, 33), 3, 8, cut=3)   # fewer than 2^cut planes per rank
    with pytest.raises(ValueError):
        SlabPlan((33, 33, 33), 2, 2, cut=3)   # cut below the coarsest level


def test_slab_forms_reduce_to_whole_array_forms(orc):
    """The oracle's slab-form operators on a whole-array 'slab' equal its original operators."""
    import ctypes as C
    from util import phi_like, rng
    g = rng(21)
    n = (17, 13, 15)
    c = orc.coeffs((0.01, 0.02, 0.03), 0)
    phi, rho = phi_like(n, g), phi_like(n, g) * orc.EPS_0
    # residual + restrict
    R, s = orc.residual(phi, rho, c)
    Rc = orc.restrict27(R)
    b = rho / orc.EPS_0
    got_Rc = np.zeros_like(Rc)
    got_s = orc.lib().orc_slab_residual_restrict(C.byref(orc.view(phi, n[2], 0)), C.byref(orc.view(b, n[2], 0)),
                                                 C.byref(orc.view(got_Rc, Rc.shape[2], 0)), (C.c_int64 * 3)(*n),
                                                 (C.c_double * 4)(*c), 0, n[0])
    assert_bit_equal(got_Rc, Rc, "slab residual+restrict")
    assert got_s == s
    # smoother (coarse form: u one node larger than rhs)
    u = phi_like(tuple(a + 1 for a in n), g)
    want = orc.smooth(u.copy(), R, n, c, 1.4, 1, rb=True)
    got = u.copy()
    orc.lib().orc_slab_smooth_rb(C.byref(orc.view(got, n[2] + 1, 0)), C.byref(orc.view(R, n[2], 0)),
                                 (C.c_int64 * 3)(*n), (C.c_double * 4)(*c), 1.4, 0, n[0] + 1)
    assert_bit_equal(got, want, "slab smoother")
    # prolongation
    ec = phi_like(tuple((a >> 1) + 1 for a in n), g)
    want = orc.prolong(ec, phi.copy(), n)
    got = phi.copy()
    orc.lib().orc_slab_prolong(C.byref(orc.view(ec, ec.shape[2], 0)), C.byref(orc.view(got, n[2], 0)),
                               (C.c_int64 * 3)(*n), 0, n[0], 0)
    assert_bit_equal(got, want, "slab prolong")


def test_library_message_sizes_match_between_neighbours_and_the_python_twin(mg):
    """What the C++ rank driver sends one way over RCCL must be what the neighbour posts a receive
    for (a mismatch is a hang on a multi-GPU node, and the in-process transport of the one-GPU tests
    cannot show it: there the receiver alone fixes the count).  mg_dist_exchange_plan gives every rank's
    counts from the configuration alone, so the whole group is checked here without a GPU: pairwise
    agreement, received planes inside the stored range and on the neighbour's own planes, and
    equality with multigrid_amd/dist.py, whose counts the gloo runs above exercise for real."""
    from multigrid_amd.dist import SlabPlan, schedule
    cases = [(1025, 6, w, 0, None) for w in (2, 3, 4, 5, 6, 7, 8)] + \
            [(2049, 7, 8, 0, None), (2049, 7, 8, 5, None), (513, 5, 8, 0, None), (513, 5, 4, 3, None), (513, 5, 3, 3, None),
             (1025, 6, 8, 4, None), (1025, 6, 8, 4, [1, 1, 9, 2, 5, 5, 50]), (1025, 6, 5, 3, [2, 7, 3, 5, 5, 5, 50]),
             (257, 4, 2, 3, [1, 3, 2, 5, 50]), (513, 5, 8, 3, None), (257, 5, 2, 0, None)]
    checked = 0
    for n0, depth, world, cut, nu in cases:
        N = (n0, n0 - 32, n0) if n0 == 513 else (n0,) * 3
        p = SlabPlan(N, depth, world, cut or None)
        nus = list(nu) if nu else list(schedule(depth))
        fits = p.deep_fits(nus)
        plans = {}
        for r in range(world):
            for l in range(1, p.cut + 1):
                plans[r, l] = mg.dist_exchange_plan(N, depth, r, world, l, coarse_cut=cut, nu=nu)
                assert plans[r, l]["deep"] == fits and plans[r, l]["cut"] == p.cut, (n0, world, cut)
                assert plans[r, l]["own"] == tuple(p.own(l, "R", r))
        off = mg.dist_exchange_plan(N, depth, 0, world, 1, coarse_cut=cut, nu=nu, deep_halos=False)
        assert not off["deep"] and off["halo"] == (0, 0, 0, 0)
        if not fits:
            assert all(v["halo"] == (0, 0, 0, 0) for v in plans.values())
            continue
        for r in range(world):
            first, last = p.deep_ranges(r, nus)
            for l in range(1, p.cut):
                me = plans[r, l]
                rlo, slo, shi, rhi = me["halo"]
                lo, hi = me["own"]
                assert me["first"] == tuple(first[l]) and me["last"] == tuple(last[l])
                # the twin's own counts (DistSolver computes them the same way from its plan)
                f = p.deep_rhs(l, first[l])
                assert (rlo, rhi) == (lo - f[0], max(f[1] - hi, 0))
                if r + 1 < world:
                    up = plans[r + 1, l]
                    assert shi == up["halo"][0] and rhi == up["halo"][1], (n0, world, l, r)
                    assert 0 <= rhi <= up["own"][1] - up["own"][0] and 0 < shi <= hi - lo
                else:
                    assert (shi, rhi) == (0, 0)
                if r == 0:
                    assert (rlo, slo) == (0, 0)
                checked += 1
            # the level that is gathered whole has nothing to exchange in this scheme
            assert plans[r, p.cut]["halo"] == (0, 0, 0, 0)
    assert checked > 100
    with pytest.raises(mg.MgError):
        mg.dist_exchange_plan((1025,) * 3, 6, 8, 8, 1)      # rank outside the world
    with pytest.raises(mg.MgError):
        mg.dist_exchange_plan((1025,) * 3, 6, 0, 8, 5)      # level below the cut
