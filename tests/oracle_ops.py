"""CPU stand-in for multigrid_amd.dist.HipOps, built on the oracle's slab-form operators.

TEST INFRASTRUCTURE: lets the distributed orchestration (partition plan, halo exchange
schedule, coarse gather/scatter) run on CPU tensors over gloo so that it can be checked
against the undivided oracle without a GPU.  Never imported by the product."""
import ctypes as C

import numpy as np
import torch

from oracle import mg_oracle as orc


class OracleOps:
    """out_of_place=True makes a sweep read u and write alt, like the GPU's single-pass kernel, so
    that the solver issues sweeps as edge and interior plane ranges around the halo exchange."""

    def __init__(self, out_of_place=False, pairs=False):
        self._oop = out_of_place
        self._pairs = pairs and out_of_place  # two sweeps per call where the stored planes allow, like k_sweep_rb2

    def sweeps_in_pairs(self, u, alt, rhs, n, lo, hi):
        def holds(s, a, b, top):
            return s.i0 <= max(a, 0) and s.i0 + s.planes >= min(b, top)
        return self._pairs and holds(u, lo - 4, hi + 4, n[0]) and holds(rhs, lo - 3, hi + 3, n[0] - 1)

    def out_of_place(self, u):
        return self._oop

    def onload_folds(self, u, correct):
        return self._oop

    def zeros(self, shape):
        return torch.zeros(shape, dtype=torch.float64)

    def upload(self, slab, array, lo):
        a = torch.from_numpy(np.ascontiguousarray(array))
        slab.rows(lo, lo + a.shape[0])[:, :, :slab.nk].copy_(a)

    def _v(self, s):
        return orc.view(s.t.numpy(), s.nk, s.i0)

    def coef(self, dh, level):
        return orc.coeffs(dh, level)

    def div_eps0(self, v):
        orc.lib().orc_slab_div_eps0(C.byref(self._v(v)))

    def smooth(self, u, alt, rhs, n, c, omega, sweeps, lo, hi, second=None):
        if second:
            assert self._oop and sweeps == 1 and second[1] - second[0] == hi - lo and second[0] >= hi
            self.smooth(u, alt, rhs, n, c, omega, 1, second[0], second[1])
        if sweeps == 2 and not second and self.sweeps_in_pairs(u, alt, rhs, n, lo, hi):
            # both sweeps in one call, the way the GPU's pair kernel does them: planes [lo,hi) of alt
            # from planes [lo-4, hi+4) of u, nothing else of alt written
            a, b = max(lo - 4, u.i0), min(hi + 4, u.i0 + u.planes)
            tmp = type(u).__new__(type(u))
            tmp.i0, tmp.nj, tmp.nk, tmp.pitch, tmp.pending = a, u.nj, u.nk, u.pitch, None
            tmp.t = u.rows(a, b).clone()
            for w in (2, 0):
                orc.lib().orc_slab_smooth_rb(C.byref(self._v(tmp)), C.byref(self._v(rhs)), (C.c_int64 * 3)(*n),
                                             (C.c_double * 4)(*c), omega, max(lo - w, a), min(hi + w, b))
            alt.rows(lo, hi).copy_(tmp.rows(lo, hi))
            return True
        if self._oop and sweeps == 1:
            # planes [lo,hi) of alt = one sweep of u: run the in-place slab form on a scratch copy of
            # the planes it touches, keep only the requested planes
            a, b = max(lo - 2, u.i0), min(hi + 2, u.i0 + u.planes)
            tmp = type(u).__new__(type(u))
            tmp.i0, tmp.nj, tmp.nk, tmp.pitch, tmp.pending = a, u.nj, u.nk, u.pitch, None
            tmp.t = u.rows(a, b).clone()
            orc.lib().orc_slab_smooth_rb(C.byref(self._v(tmp)), C.byref(self._v(rhs)), (C.c_int64 * 3)(*n),
                                         (C.c_double * 4)(*c), omega, lo, hi)
            alt.rows(lo, hi).copy_(tmp.rows(lo, hi))
            return True
        for _ in range(sweeps):
            orc.lib().orc_slab_smooth_rb(C.byref(self._v(u)), C.byref(self._v(rhs)), (C.c_int64 * 3)(*n),
                                         (C.c_double * 4)(*c), omega, lo, hi)
        return False  # in place

    def prolong_smooth(self, ec, correct, u, alt, rhs, n, c, omega, lo, hi, second=None):
        if second:
            assert self._oop
            self.prolong_smooth(ec, correct, u, alt, rhs, n, c, omega, second[0], second[1])
        plo, phi_ = max(lo - 2, u.i0), min(hi + 2, u.i0 + u.planes, n[0])
        if self._oop:
            # on-load form: u itself stays as it was, the sweep sees the prolongated values
            src = type(u).__new__(type(u))
            src.i0, src.nj, src.nk, src.pitch, src.pending = u.i0, u.nj, u.nk, u.pitch, None
            src.t = u.t.clone()
            self.prolong(ec, src, n, plo, phi_, correct)
            return self.smooth(src, alt, rhs, n, c, omega, 1, lo, hi)
        self.prolong(ec, u, n, plo, phi_, correct)
        return self.smooth(u, alt, rhs, n, c, omega, 1, lo, hi)

    def residual_restrict(self, phi, b, Rc, n, c, parts, before_part=None):
        s = 0.0
        for idx, (lo, hi) in enumerate(parts):
            if before_part:
                before_part(idx)
            s += orc.lib().orc_slab_residual_restrict(C.byref(self._v(phi)), C.byref(self._v(b)),
                                                      C.byref(self._v(Rc)), (C.c_int64 * 3)(*n),
                                                      (C.c_double * 4)(*c), lo, hi)
        return torch.tensor([s], dtype=torch.float64)

    def restrict27(self, Rf, Rc, nf, lo, hi):
        orc.lib().orc_slab_restrict27(C.byref(self._v(Rf)), C.byref(self._v(Rc)), (C.c_int64 * 3)(*nf), lo, hi)

    def prolong(self, ec, out, n, lo, hi, correct):
        orc.lib().orc_slab_prolong(C.byref(self._v(ec)), C.byref(self._v(out)), (C.c_int64 * 3)(*n), lo, hi,
                                   int(correct))

    def sync(self):
        pass
