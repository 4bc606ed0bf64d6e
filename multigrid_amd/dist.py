"""Slab-partitioned multigrid V-cycle: one process per GPU, halo planes over
torch.distributed (backend "nccl" = RCCL over xGMI on a GPU node; "gloo" works too).

The reference is a single-threaded CPU program; nothing here has a counterpart in it.
What is distributed is the same cycle the single-GPU library runs
(reference cpp/PotentialSolver.cpp:3756-4216), operator by operator, with results
bit-identical to the undivided red-black cycle:

  * the i axis (slowest, planes contiguous) of every level l < cut is split into slabs
    whose boundaries are multiples of 2^cut fine planes, so coarse slabs nest;
  * every rank stores its planes plus halo planes (2 for phi/eps, 1 for right-hand
    sides).  A red-black sweep recomputes the red update of the first halo plane on
    either side, so ONE exchange of two planes per side per sweep replaces an exchange
    after every colour;
  * where the sweep kernel is out of place, a sweep is issued as three plane ranges: the two
    planes next to each neighbour first, then the exchange of exactly those planes is started,
    then the interior -- which needs no halo plane of the previous sweep either -- so the
    exchange runs under the interior sweep; the fine residual is split the same way;
  * deep halos on the coarse distributed levels: where every rank finds them on its two
    neighbours, a rank receives once per cycle as many planes of R_l as all nu_l sweeps of level l
    reach (two per sweep), forms eps_l on its slab widened by that much and sweeps ranges that
    shrink by two planes per side and sweep, with no exchange between the sweeps;
  * the sum of squared residuals is all-reduced once per cycle;
  * R of level `cut` is all-gathered and every rank solves level `cut` and everything below it
    on whole arrays of its own (those levels are serial work of the cycle whoever runs them);
    eps_cut then needs no transfer: each rank prolongs from its own copy.

The orchestration below is backend-agnostic: `ops` runs one operator on local slabs
(HipOps: the mg_dev_* entry points of libmg_hip on device tensors).  There is no CPU
compute backend in the product; the test-suite plugs in its CPU oracle to check the
partition and the exchange schedule.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.distributed as dist


# --------------------------------------------------------------------------- plan

class SlabPlan:
    """Who owns which planes of which level.  Pure arithmetic; no device, no comm."""

    def __init__(self, n0, depth, world, cut=None, cut_size=128):
        self.n0 = tuple(int(a) for a in n0)
        self.depth = int(depth)
        self.world = int(world)
        L = self.depth
        if (min(self.n0) >> L) < 3:
            raise ValueError("grid too small for this depth")
        if cut is None:  # first level small enough to live on one GPU, at least 1
            cut = next((l for l in range(1, L + 1) if (self.n0[0] >> l) <= cut_size), L)
        if not 1 <= cut <= L:
            raise ValueError("cut level must be in 1..depth")
        self.cut = cut
        align = 1 << cut
        units = (self.n0[0] - 1) // align  # slabs are whole units of 2^cut fine planes
        even = units // self.world
        if even < 1:
            raise ValueError(f"{self.n0[0]} planes over {self.world} ranks leave fewer than "
                             f"2^{cut} fine planes per rank; lower the cut level or the rank count")
        W = self.world
        self.bounds = [0] * W + [self.n0[0]]  # fine-plane boundaries
        if W == 1:
            return
        # whole units as evenly as they go, the spare ones to the lower ranks (as Plan::init in
        # csrc/mg_dist.hip)
        spare = units % W
        for r in range(1, W):
            self.bounds[r] = self.bounds[r - 1] + (even + (1 if r - 1 < spare else 0)) * align

    def level_n(self, l):
        return tuple(a >> l for a in self.n0)

    def planes(self, l, kind):
        """Global plane count of an array of level l: 'eps' arrays of l >= 1 have one extra node."""
        n = self.n0[0] >> l
        return n + 1 if (kind == "eps" and l >= 1) else n

    def dims(self, l, kind):
        n = self.level_n(l)
        return tuple(a + 1 for a in n) if (kind == "eps" and l >= 1) else n

    def own(self, l, kind, rank):
        """Planes [lo, hi) of level l that `rank` owns (the last rank takes the tail)."""
        lo = self.bounds[rank] >> l
        hi = self.planes(l, kind) if rank == self.world - 1 else self.bounds[rank + 1] >> l
        return lo, hi

    def stored(self, l, kind, rank, halo):
        """Planes [lo, hi) a rank stores: owned plus `halo` planes per side, clipped to the array."""
        lo, hi = self.own(l, kind, rank)
        return max(lo - halo, 0), min(hi + halo, self.planes(l, kind))

    # -- deep halos (as Plan::deep_ranges / deep_fits in csrc/mg_dist.hip)
    def deep_ranges(self, rank, nu, halo=2):
        """Per level l in 1..cut-1: (first, last) = the planes of eps_l a rank forms by prolongation
        before the level's first sweep and must hold after its last.  A sweep needs its input two
        planes further out on either side than it produces, so sweep s of nu produces `last` widened
        by 2*(nu-s); the level below is needed on the planes the prolongation onto `first` reads."""
        need = self.stored(0, "phi", rank, halo)  # fine planes the correction phi -= P(eps_1) reaches
        first, last = {}, {}
        for l in range(1, self.cut):
            top = self.planes(l, "eps")
            last[l] = (need[0] >> 1, min((need[1] >> 1) + 1, top))
            first[l] = (max(last[l][0] - 2 * nu[l], 0), min(last[l][1] + 2 * nu[l], top))
            need = first[l]
        return first, last

    def deep_rhs(self, l, first):
        """The planes of R_l the sweeps on `first` read: those of eps_l, as far as R_l has them."""
        return first[0], min(first[1], self.planes(l, "R"))

    def deep_fits(self, nu):
        """Deep halos are used when every rank finds the planes beyond its own on its two neighbours."""
        if self.world < 2 or self.cut < 2 or any(nu[l] > 4096 for l in range(1, self.cut)):
            return False
        for r in range(self.world):
            first, _ = self.deep_ranges(r, nu)
            for l in range(1, self.cut):
                lo = self.own(l, "eps", r - 1)[0] if r > 0 else 0
                hi = self.own(l, "eps", r + 1)[1] if r + 1 < self.world else self.planes(l, "eps")
                if first[l][0] < lo or first[l][1] > hi:
                    return False
        return True


# --------------------------------------------------------------------------- storage

class Slab:
    """Planes [i0, i0+planes) of a level's [i][j][k] array as a (planes, nj, pitch) fp64 tensor;
    pitch = nk rounded up to even (the device library's row pitch)."""

    def __init__(self, ops, i0, planes, nj, nk):
        self.i0, self.nj, self.nk = int(i0), int(nj), int(nk)
        self.pitch = (self.nk + 1) & ~1
        self.t = ops.zeros((int(planes), self.nj, self.pitch))
        self.pending = None  # halo exchange in flight into this slab (Comm.exchange_start)

    def ready(self):
        """Block the compute stream until the halo planes in flight have arrived."""
        if self.pending is not None:
            self.pending()
            self.pending = None

    @property
    def planes(self):
        return self.t.shape[0]

    def rows(self, lo, hi):
        """Tensor view of global planes [lo, hi)."""
        return self.t[lo - self.i0:hi - self.i0]

    def data(self, lo=None, hi=None):
        """numpy copy of planes [lo,hi) without the pitch padding (tests / output)."""
        lo = self.i0 if lo is None else lo
        hi = self.i0 + self.planes if hi is None else hi
        return self.rows(lo, hi)[:, :, :self.nk].cpu().numpy().copy()


class HipOps:
    """Local operators on device slabs through libmg_hip's mg_dev_* entry points."""

    def __init__(self, device_index):
        from . import libmg  # torch is already imported: both share one HIP runtime
        self.libmg = libmg
        self.lib = libmg.load()
        if not torch.cuda.is_available():
            raise RuntimeError("HipOps needs a GPU; there is no CPU fallback")
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.partials = torch.zeros(1 << 16, dtype=torch.float64, device=self.device)
        self.scalar = torch.zeros(1, dtype=torch.float64, device=self.device)

    # -- storage
    def zeros(self, shape):
        return torch.zeros(shape, dtype=torch.float64, device=self.device)

    def upload(self, slab, array, lo):
        """Dense host array of planes [lo, lo+len) -> slab."""
        a = torch.from_numpy(np.ascontiguousarray(array))
        slab.rows(lo, lo + a.shape[0])[:, :, :slab.nk].copy_(a)

    # -- helpers
    def _arr(self, s):
        return self.libmg.MgDevArray(s.t.data_ptr(), s.planes, s.nj, s.nk, s.i0)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        if rc != 0:
            raise self.libmg.MgError(rc, self.lib.mg_last_error().decode(errors="replace"))

    def coef(self, dh, level):
        out = (C.c_double * 4)()
        self._check(self.lib.mg_level_coef((C.c_double * 3)(*dh), level, out))
        return tuple(out)

    # -- operators
    def out_of_place(self, u):
        """True when a sweep of this slab reads u and writes alt (so plane ranges are independent)."""
        return bool(self.lib.mg_dev_sweep_out_of_place(C.byref(self._arr(u))))

    def div_eps0(self, v):
        self._check(self.lib.mg_dev_div_eps0(self._stream(), C.byref(self._arr(v))))

    def onload_folds(self, u, correct):
        """True when prolong_smooth leaves u untouched (the operation rides on the sweep's loads)."""
        return bool(self.lib.mg_dev_onload_folds(C.byref(self._arr(u)), int(correct)))

    def sweeps_in_pairs(self, u, alt, rhs, n, lo, hi):
        """True when smooth(..., sweeps=2, lo, hi) does both sweeps in one launch (reading planes
        lo-4 .. hi+3 of u), so that a caller narrowing its range from sweep to sweep may hand over two."""
        return bool(self.lib.mg_dev_sweeps_in_pairs(C.byref(self._arr(u)), C.byref(self._arr(rhs)),
                                                    (C.c_int64 * 3)(*n), lo, hi))

    def smooth(self, u, alt, rhs, n, c, omega, sweeps, lo, hi, second=None):
        """Returns True when the result landed in `alt`.  second = (lo2, hi2): another range of
        the same length swept in the same launch (out-of-place kernel, one sweep)."""
        flag = C.c_int(0)
        lo2, hi2 = second or (0, 0)
        self._check(self.lib.mg_dev_smooth_rb(self._stream(), C.byref(self._arr(u)), C.byref(self._arr(alt)),
                                              C.byref(self._arr(rhs)), (C.c_int64 * 3)(*n), (C.c_double * 4)(*c),
                                              omega, sweeps, lo, hi, lo2, hi2, C.byref(flag)))
        return bool(flag.value)

    def prolong_smooth(self, ec, correct, u, alt, rhs, n, c, omega, lo, hi, second=None):
        """One sweep of planes [lo,hi) (and `second`) preceded by u := P(ec) (or u -= P(ec) when
        correct) on the stored planes the sweep reads.  Returns True when the result landed in `alt`."""
        flag = C.c_int(0)
        lo2, hi2 = second or (0, 0)
        self._check(self.lib.mg_dev_prolong_smooth_rb(
            self._stream(), C.byref(self._arr(ec)), int(correct), C.byref(self._arr(u)), C.byref(self._arr(alt)),
            C.byref(self._arr(rhs)), (C.c_int64 * 3)(*n), (C.c_double * 4)(*c), omega, 1, lo, hi, lo2, hi2,
            C.byref(flag)))
        return bool(flag.value)

    def residual_restrict(self, phi, b, Rc, n, c, parts, before_part=None):
        """Fused residual + restriction over the plane ranges `parts` (each even-aligned), one launch
        per range; before_part(i) runs ahead of launch i.  Returns a 1-element tensor holding this
        rank's sum of squared residuals."""
        used = 0
        for idx, (lo, hi) in enumerate(parts):
            if before_part:
                before_part(idx)
            count = C.c_int64(0)
            self._check(self.lib.mg_dev_residual_restrict(
                self._stream(), C.byref(self._arr(phi)), C.byref(self._arr(b)), C.byref(self._arr(Rc)),
                (C.c_int64 * 3)(*n), (C.c_double * 4)(*c), C.c_void_p(self.partials.data_ptr() + 8 * used),
                self.partials.numel() - used, lo, hi, C.byref(count)))
            used += count.value
        self._check(self.lib.mg_dev_sum_partials(self._stream(), C.c_void_p(self.partials.data_ptr()),
                                                 used, C.c_void_p(self.scalar.data_ptr())))
        return self.scalar

    def restrict27(self, Rf, Rc, nf, lo, hi):
        self._check(self.lib.mg_dev_restrict27(self._stream(), C.byref(self._arr(Rf)), C.byref(self._arr(Rc)),
                                               (C.c_int64 * 3)(*nf), lo, hi))

    def prolong(self, ec, out, n, lo, hi, correct):
        self._check(self.lib.mg_dev_prolong(self._stream(), C.byref(self._arr(ec)), C.byref(self._arr(out)),
                                            (C.c_int64 * 3)(*n), lo, hi, int(correct)))

    def sync(self):
        torch.cuda.synchronize(self.device)


# --------------------------------------------------------------------------- communication

class Comm:
    """Halo exchange, all-reduce and gather/scatter over torch.distributed point-to-point ops."""

    def __init__(self, group=None, stage_through_host=None, defer=False):
        self.group = group
        # defer: run a started exchange only when its slab is next needed (slab.ready()).  RCCL moves
        # the planes some time between start and completion; the blocking path is the earliest case,
        # this is the latest one -- the tests run both to pin the schedule's dependencies.
        self.defer = defer
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bytes_sent = 0
        # gloo moves only host tensors point to point: device planes are staged through the host
        if stage_through_host is None:
            stage_through_host = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self.stage = stage_through_host

    def _run(self, ops):
        if not ops:
            return
        if not self.stage or not any(o.tensor.is_cuda for o in ops):
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            return
        staged, back = [], []
        for o in ops:
            h = o.tensor.cpu() if o.op is dist.isend else torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
            staged.append(dist.P2POp(o.op, h, o.peer, self.group))
            if o.op is dist.irecv:
                back.append((o.tensor, h))
        for w in dist.batch_isend_irecv(staged):
            w.wait()
        for dst, h in back:
            dst.copy_(h)

    def _halo_ops(self, slab, own, width):
        """width: planes per side, or (rlo, slo, shi, rhi) = planes received below the owned range,
        sent from its low end, sent from its high end, received above it (what a rank sends is what
        its neighbour receives; a side with nothing to move is left out on both ranks)."""
        lo, hi = own
        rlo, slo, shi, rhi = (width,) * 4 if isinstance(width, int) else width
        ops = []
        if self.rank + 1 < self.world:
            assert slab.i0 + slab.planes - hi >= rhi  # halo planes stored above
            if shi > 0:
                ops.append(dist.P2POp(dist.isend, slab.rows(hi - shi, hi), self.rank + 1, self.group))
            if rhi > 0:
                ops.append(dist.P2POp(dist.irecv, slab.rows(hi, hi + rhi), self.rank + 1, self.group))
        if self.rank > 0:
            if slo > 0:
                ops.append(dist.P2POp(dist.isend, slab.rows(lo, lo + slo), self.rank - 1, self.group))
            if rlo > 0:
                ops.append(dist.P2POp(dist.irecv, slab.rows(lo - rlo, lo), self.rank - 1, self.group))
        self.bytes_sent += sum(o.tensor.numel() * 8 for o in ops if o.op is dist.isend)
        return ops

    def exchange(self, slab, own, width):
        """Fill this rank's halo planes of `slab` (`width` per side) from the neighbours' owned
        planes and send ours: planes are contiguous, one message per neighbour per direction."""
        if self.world == 1:
            return
        slab.ready()
        self._run(self._halo_ops(slab, own, width))

    def exchange_start(self, slab, own, width):
        """The same exchange, started now and completed by slab.ready().  With RCCL the transfers
        are ordered after the kernels enqueued so far and run beside the ones enqueued next; the
        host-staged path (gloo) completes here."""
        if self.world == 1:
            return
        slab.ready()
        ops = self._halo_ops(slab, own, width)
        if not ops:
            return
        if self.defer:
            slab.pending = lambda: self._run(ops)
            return
        if self.stage and any(o.tensor.is_cuda for o in ops) or not ops[0].tensor.is_cuda:
            self._run(ops)
            return
        works = dist.batch_isend_irecv(ops)

        def finish():
            for w in works:
                w.wait()
        slab.pending = finish

    def allreduce_sum(self, t):
        if self.world > 1:
            if self.stage and t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def allgather_planes(self, plan, l, kind, slab, full):
        """Owned planes of every rank -> every rank's whole array `full`."""
        lo, hi = plan.own(l, kind, self.rank)
        full.rows(lo, hi).copy_(slab.rows(lo, hi))
        ops = []
        for r in range(self.world):
            if r == self.rank:
                continue
            olo, ohi = plan.own(l, kind, r)
            ops.append(dist.P2POp(dist.isend, slab.rows(lo, hi), r, self.group))
            ops.append(dist.P2POp(dist.irecv, full.rows(olo, ohi), r, self.group))
        self.bytes_sent += sum(o.tensor.numel() * 8 for o in ops if o.op is dist.isend)
        self._run(ops)


# --------------------------------------------------------------------------- solver

REFERENCE_SCHEDULES = {1: (3, 50), 2: (3, 5, 50), 3: (3, 5, 5, 50), 4: (3, 5, 5, 5, 50),
                       5: (1, 5, 5, 5, 5, 50)}  # reference cpp/PotentialSolver.cpp:2422.. :3759-3770


def schedule(depth):
    return REFERENCE_SCHEDULES.get(depth) or (1,) + (5,) * (depth - 1) + (50,)


class DistSolver:
    """One rank's share of a slab-partitioned multigrid solver (same state as a reference solver
    object, divided).  Construct on every rank of the group with the same arguments."""

    H_U, H_RHS = 2, 1  # halo planes of swept fields / of right-hand sides

    def __init__(self, n0, dh, depth, ops, comm, nu=None, omega=1.4, cut=None, cut_size=128, overlap=True,
                 deep_halos=True):
        self.ops, self.comm = ops, comm
        self.overlap = overlap  # run halo exchanges under the interior part of the sweeps
        self.rank, self.world = comm.rank, comm.world
        self.plan = SlabPlan(n0, depth, self.world, cut, cut_size)
        self.depth, self.cut = depth, self.plan.cut
        self.dh = tuple(float(a) for a in dh)
        self.nu = tuple(nu) if nu is not None else schedule(depth)
        self.omega = omega
        self.coef = [ops.coef(self.dh, l) for l in range(depth + 1)]
        P, r = self.plan, self.rank
        # deep halos on levels 1..cut-1: eps_l and R_l reach as far beyond the owned planes as all
        # sweeps of the level do, and R_l's planes there arrive once per cycle
        self.deep = bool(deep_halos) and P.deep_fits(self.nu)
        if self.deep:
            self.dfirst, self.dlast = P.deep_ranges(r, self.nu)
            below = P.deep_ranges(r - 1, self.nu)[0] if r > 0 else None
            above = P.deep_ranges(r + 1, self.nu)[0] if r + 1 < self.world else None
            self.dhalo = {}
            for l in range(1, self.cut):
                lo, hi = P.own(l, "R", r)
                f = P.deep_rhs(l, self.dfirst[l])
                self.dhalo[l] = (lo - f[0], max(P.deep_rhs(l, below[l])[1] - lo, 0) if below else 0,
                                 hi - P.deep_rhs(l, above[l])[0] if above else 0, max(f[1] - hi, 0))

        def slab(l, kind, halo):
            lo, hi = P.stored(l, kind, r, halo)
            if self.deep and 1 <= l < self.cut:
                lo, hi = self.dfirst[l] if kind == "eps" else P.deep_rhs(l, self.dfirst[l])
            d = P.dims(l, kind)
            return Slab(ops, lo, hi - lo, d[1], d[2])

        def whole(l, kind):
            d = P.dims(l, kind)
            return Slab(ops, 0, d[0], d[1], d[2])

        # distributed levels 0..cut-1 (phi/eps with a ping-pong twin), plus R and eps of level `cut`
        self.phi, self.phi_alt = slab(0, "phi", self.H_U), slab(0, "phi", self.H_U)
        self.b = slab(0, "R", self.H_RHS)
        self.R = {l: slab(l, "R", self.H_RHS) for l in range(1, self.cut + 1)}
        self.eps = {l: slab(l, "eps", self.H_U) for l in range(1, self.cut + 1)}
        self.eps_alt = {l: slab(l, "eps", self.H_U) for l in range(1, self.cut)}
        # levels cut..depth: whole arrays on every rank
        self.cR, self.ceps, self.ceps_alt = {}, {}, {}
        for l in range(self.cut, depth + 1):
            self.cR[l], self.ceps[l], self.ceps_alt[l] = whole(l, "R"), whole(l, "eps"), whole(l, "eps")
        self.n_nodes = float(self.plan.n0[0]) * self.plan.n0[1] * self.plan.n0[2]
        self.last_sum = None
        self.correction_pending = False

    # -- data in / out
    def own(self, l, kind):
        return self.plan.own(l, kind, self.rank)

    UPLOAD_PLANES = 32  # host arrays are produced and uploaded this many planes at a time

    def _upload(self, slab, of_planes, lo, hi):
        for a in range(lo, hi, self.UPLOAD_PLANES):
            b = min(a + self.UPLOAD_PLANES, hi)
            self.ops.upload(slab, of_planes(a, b), a)

    def set_rho(self, rho_of_planes):
        """rho_of_planes(lo, hi) -> dense array of global planes [lo,hi); called for a few planes at
        a time, so a rank never holds its whole slab in host memory.  Stored as rho/EPS_0."""
        lo, hi = self.plan.stored(0, "R", self.rank, self.H_RHS)
        self._upload(self.b, rho_of_planes, lo, hi)
        self.ops.div_eps0(self.b)

    def set_phi(self, phi_of_planes):
        lo, hi = self.plan.stored(0, "phi", self.rank, self.H_U)
        self._upload(self.phi, phi_of_planes, lo, hi)
        self.correction_pending = False

    def get_phi_owned(self):
        self.flush_correction()
        self.phi.ready()
        lo, hi = self.own(0, "phi")
        return lo, self.phi.data(lo, hi)

    # -- the cycle
    def _split(self, own, first, last):
        """Plane ranges of an owned slab: the two planes next to each existing neighbour (what a
        halo exchange of width 2 sends) and the interior between them."""
        lo, hi = own
        a = lo + self.H_U if not first else lo
        b = hi - self.H_U if not last else hi
        return ([(lo, a)] if a > lo else []) + ([(b, hi)] if b < hi else []), (a, b)

    def _sweeps(self, l, ec=None, correct=False):
        """nu[l] red-black sweeps of a distributed level, one halo exchange after each.  With `ec`
        (eps of level l+1, halos current) the first sweep is preceded by the prolongation from it:
        u := P(ec), or u -= P(ec) when `correct` -- folded into that sweep's loads where possible."""
        n = self.plan.level_n(l)
        first, last = self.rank == 0, self.rank == self.world - 1
        for sweep in range(self.nu[l]):
            if l == 0:
                u, alt, rhs, own = self.phi, self.phi_alt, self.b, self.own(0, "phi")
            else:
                u, alt, rhs, own = self.eps[l], self.eps_alt[l], self.R[l], self.own(l, "eps")

            fold = None
            if sweep == 0 and ec is not None:
                ec.ready()
                if self.ops.onload_folds(u, correct):
                    fold = ec  # the operation rides on this sweep's loads; u stays as it is
                else:
                    # applied in place once, on every stored plane this sweep reads
                    u.ready()
                    plo, phi_ = max(own[0] - self.H_U, u.i0), min(own[1] + self.H_U, u.i0 + u.planes)
                    self.ops.prolong(ec, u, n, plo, phi_, correct)

            def run(lo, hi, second=None):
                if fold is not None:
                    return self.ops.prolong_smooth(fold, correct, u, alt, rhs, n, self.coef[l], self.omega,
                                                   lo, hi, second)
                return self.ops.smooth(u, alt, rhs, n, self.coef[l], self.omega, 1, lo, hi, second)

            edges, inner = self._split(own, first, last)
            overlap = (self.world > 1 and self.overlap and self.ops.out_of_place(u)
                       and inner[1] - inner[0] >= 2 * self.H_U)
            # The last fine sweep is followed by the residual pass, whose interior part reads no
            # halo plane: the sweep runs as ONE launch and its exchange under that interior part
            # (an edge launch re-reads four lead-in planes per edge).
            whole_then_exchange = overlap and l == 0 and sweep == self.nu[l] - 1
            if whole_then_exchange:
                u.ready()
                assert run(*own), "out-of-place sweep expected"
                self.comm.exchange_start(alt, own, self.H_U)
                self.phi, self.phi_alt = alt, u
                continue
            if overlap:
                # edge planes need the halos of the previous sweep; the interior does not.  Both
                # edges (equally long) go in one launch.
                u.ready()
                swapped = [run(*edges[0], edges[1] if len(edges) == 2 else None)]
                self.comm.exchange_start(alt, own, self.H_U)
                swapped.append(run(*inner))
                assert all(swapped), "out-of-place sweep expected"
                swapped = True
            else:
                u.ready()
                swapped = run(*own)
            if swapped:
                if l == 0:
                    self.phi, self.phi_alt = alt, u
                else:
                    self.eps[l], self.eps_alt[l] = alt, u
                u = alt
            if not overlap:
                self.comm.exchange(u, own, self.H_U)

    def flush_correction(self):
        """Apply a coarse-grid correction that is still waiting for the next fine sweep."""
        if self.correction_pending:
            # phi -= P(eps_1) on every stored plane, halos included: eps_1's halos are current
            lo, hi = self.plan.stored(0, "phi", self.rank, self.H_U)
            self.eps[1].ready()
            self.phi.ready()
            self.ops.prolong(self.eps[1], self.phi, self.plan.n0, lo, hi, True)
            self.correction_pending = False

    def smooth_and_residual(self):
        """Steps 1)-2b) of the reference iteration (+ the restriction to level 1, which shares the
        pass).  Returns the global sum of squared residuals as a 1-element tensor."""
        if self.correction_pending and self.nu[0] > 0:
            self._sweeps(0, self.eps[1], True)   # step 6) of the previous iteration rides along
            self.correction_pending = False
        else:
            self.flush_correction()
            self._sweeps(0)
        own = self.own(0, "phi")
        edges, inner = self._split(own, self.rank == 0, self.rank == self.world - 1)
        # slab boundaries are multiples of 2^cut, so every range below starts on an even plane
        if self.world > 1 and self.overlap and inner[1] - inner[0] >= 2 * self.H_U:
            # the interior needs no halo plane: it runs while the last sweep's exchange completes
            parts = [inner] + ([(own[0], inner[0])] if inner[0] > own[0] else []) + \
                    ([(inner[1], own[1])] if inner[1] < own[1] else [])
            wait_at = 1
        else:
            parts, wait_at = [own], 0
        s = self.ops.residual_restrict(self.phi, self.b, self.R[1], self.plan.n0, self.coef[0], parts,
                                       lambda idx: self.phi.ready() if idx == wait_at else None)
        self.phi.ready()
        self.last_sum = self.comm.allreduce_sum(s)
        return self.last_sum

    def _sweeps_deep(self, l, below):
        """The nu[l] sweeps of a distributed level on deep halos: eps_l = P(eps_{l+1}) on dfirst[l], then
        sweep s on dlast[l] widened by 2*(nu-s) planes per side.  No exchange: the planes beyond the
        owned ones are computed here as the neighbour computes them, from R_l planes received once."""
        P, ops = self.plan, self.ops
        n, nu, top = P.level_n(l), self.nu[l], P.planes(l, "eps")
        u, alt, rhs = self.eps[l], self.eps_alt[l], self.R[l]
        below.ready()
        rhs.ready()
        last = self.dlast[l]

        def out(s):
            w = 2 * (nu - s)
            return max(last[0] - w, 0), min(last[1] + w, top)

        s = 1
        if nu > 0 and ops.onload_folds(u, False):
            assert ops.prolong_smooth(below, False, u, alt, rhs, n, self.coef[l], self.omega, *out(1)), \
                "out-of-place sweep expected"
            u, alt = alt, u
            s = 2
        else:
            f = self.dfirst[l]
            ops.prolong(below, u, n, f[0], min(f[1], n[0]), False)
        while s <= nu:
            if s + 1 <= nu and ops.sweeps_in_pairs(u, alt, rhs, n, *out(s + 1)):
                swapped = ops.smooth(u, alt, rhs, n, self.coef[l], self.omega, 2, *out(s + 1))
                s += 2
            else:
                swapped = ops.smooth(u, alt, rhs, n, self.coef[l], self.omega, 1, *out(s))
                s += 1
            if swapped:
                u, alt = alt, u
        self.eps[l], self.eps_alt[l] = u, alt

    def coarse_correction(self):
        """Steps 3)-5): restrict down to the cut level, all-gather it, solve the small levels (every
        rank), then (prolong, smooth) up.  Step 6), the correction of phi, is left pending for the
        next iteration's first sweep (flush_correction() applies it when no iteration follows)."""
        P, ops, comm = self.plan, self.ops, self.comm
        for l in range(1, self.cut):
            # R_l's halo plane below feeds the restriction; both feed the sweeps' red halo planes
            # (with deep halos the plane above comes later, with the others the sweeps reach)
            comm.exchange(self.R[l], self.own(l, "R"), (self.H_RHS, 0, self.H_RHS, 0) if self.deep else self.H_RHS)
            lo, hi = self.own(l + 1, "R")
            ops.restrict27(self.R[l], self.R[l + 1], P.level_n(l), lo, hi)
        comm.allgather_planes(P, self.cut, "R", self.R[self.cut], self.cR[self.cut])
        if self.deep:
            # the planes of R_l the sweeps of level l reach beyond the owned ones: one message per side
            # and level, in the order they are needed; they travel while the whole levels are solved
            for l in range(self.cut - 1, 0, -1):
                comm.exchange_start(self.R[l], self.own(l, "R"), self.dhalo[l])
            self._coarse_levels()
            for l in range(self.cut - 1, 0, -1):
                self._sweeps_deep(l, self.ceps[self.cut] if l + 1 == self.cut else self.eps[l + 1])
            self.correction_pending = True
            return
        self._coarse_levels()
        # eps_cut on this rank's stored planes: a local copy out of its own whole array
        lo, hi = P.stored(self.cut, "eps", self.rank, self.H_U)
        self.eps[self.cut].rows(lo, hi).copy_(self.ceps[self.cut].rows(lo, hi))
        for l in range(self.cut - 1, 0, -1):
            if self.nu[l] > 0:
                # eps_l = P(eps_{l+1}) as the first sweep's input, on halo planes too (eps_{l+1}'s
                # halos are current), so the prolongation needs no exchange of its own
                self._sweeps(l, self.eps[l + 1], False)
            else:
                lo, hi = P.stored(l, "eps", self.rank, self.H_U)
                self.eps[l + 1].ready()
                ops.prolong(self.eps[l + 1], self.eps[l], P.level_n(l), lo, hi, False)
        # step 6), phi -= P(eps_1), waits for the next iteration's first fine sweep
        self.correction_pending = True

    def _coarse_ranges(self):
        """Per whole level l: (first, last) = the planes of eps_l that must be right before its first
        and after its last sweep.  A rank reads of eps_cut only the planes under its own slab of
        level cut-1, and a sweep needs its input two planes further out on either side than it
        produces; a level is cut down this way when that leaves at most half of it."""
        P, L, H = self.plan, self.depth, 2
        need = self.dfirst[self.cut - 1] if self.deep else P.stored(self.cut - 1, "eps" if self.cut > 1 else "phi", self.rank, H)
        first, last = {}, {}
        for l in range(self.cut, L + 1):
            top = P.planes(l, "eps")
            last[l] = (need[0] >> 1, min((need[1] >> 1) + 1, top))
            first[l] = (max(last[l][0] - 2 * self.nu[l], 0), min(last[l][1] + 2 * self.nu[l], top))
            if l == L or self.world == 1 or 2 * (first[l][1] - first[l][0]) > top:
                first[l] = last[l] = (0, top)
            need = first[l]
        return first, last

    def _coarse_levels(self):
        """Levels cut..depth on whole arrays (every rank its own): restrict down, smooth the coarsest,
        (prolong, smooth) back up to level `cut`, each level on the planes `_coarse_ranges` names."""
        P, ops, L = self.plan, self.ops, self.depth
        for l in range(self.cut, L):
            ops.restrict27(self.cR[l], self.cR[l + 1], P.level_n(l), 0, P.planes(l + 1, "R"))
        first, last = self._coarse_ranges()
        for l in range(L, self.cut - 1, -1):
            top = P.planes(l, "eps")
            if l < L:
                ops.prolong(self.ceps[l + 1], self.ceps[l], P.level_n(l), first[l][0], min(first[l][1], P.planes(l, "R")), False)
            u, alt = self.ceps[l], self.ceps_alt[l]
            if first[l] == (0, top):
                if ops.smooth(u, alt, self.cR[l], P.level_n(l), self.coef[l], self.omega, self.nu[l], 0, top):
                    u, alt = alt, u
            else:
                for s in range(1, self.nu[l] + 1):
                    w = 2 * (self.nu[l] - s)
                    if ops.smooth(u, alt, self.cR[l], P.level_n(l), self.coef[l], self.omega, 1, max(last[l][0] - w, 0), min(last[l][1] + w, top)):
                        u, alt = alt, u
            self.ceps[l], self.ceps_alt[l] = u, alt

    def l2(self, s=None):
        s = self.last_sum if s is None else s
        return math.sqrt(float(s.item()) / self.n_nodes)

    def vcycles(self, count):
        """`count` full iterations (no termination test); returns the last residual norm."""
        for _ in range(count):
            s = self.smooth_and_residual()
            self.coarse_correction()
        self.flush_correction()
        self.ops.sync()
        return self.l2(s)

    def solve(self, max_it=10000, tol=1e-4):
        """The reference iteration loop: stop in the first iteration whose norm is < tol, before
        that iteration's correction.  Returns (converged, residual evaluations, norm)."""
        l2 = 0.0
        for it in range(max_it):
            l2 = self.l2(self.smooth_and_residual())
            if l2 < tol:
                return True, it + 1, l2
            self.coarse_correction()
        self.flush_correction()
        return False, max_it, l2
