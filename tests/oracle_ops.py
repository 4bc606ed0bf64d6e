"""CPU stand-in for multigrid_amd.dist.HipOps, built on the oracle's slab-form operators.

TEST INFRASTRUCTURE: lets the distributed orchestration (partition plan, halo exchange
schedule, coarse gather/scatter) run on CPU tensors over gloo so that it can be checked
against the undivided oracle without a GPU.  Never imported by the product."""
import ctypes as C

import numpy as np
import torch

from oracle import mg_oracle as orc


class OracleOps:
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

    def smooth(self, u, alt, rhs, n, c, omega, sweeps, lo, hi):
        for _ in range(sweeps):
            orc.lib().orc_slab_smooth_rb(C.byref(self._v(u)), C.byref(self._v(rhs)), (C.c_int64 * 3)(*n),
                                         (C.c_double * 4)(*c), omega, lo, hi)
        return False  # in place

    def prolong_smooth(self, ec, correct, u, alt, rhs, n, c, omega, lo, hi):
        plo, phi_ = max(lo - 2, u.i0), min(hi + 2, u.i0 + u.planes, n[0])
        self.prolong(ec, u, n, plo, phi_, correct)
        return self.smooth(u, alt, rhs, n, c, omega, 1, lo, hi)

    def residual_restrict(self, phi, b, Rc, n, c, lo, hi):
        s = orc.lib().orc_slab_residual_restrict(C.byref(self._v(phi)), C.byref(self._v(b)), C.byref(self._v(Rc)),
                                                 (C.c_int64 * 3)(*n), (C.c_double * 4)(*c), lo, hi)
        return torch.tensor([s], dtype=torch.float64)

    def restrict27(self, Rf, Rc, nf, lo, hi):
        orc.lib().orc_slab_restrict27(C.byref(self._v(Rf)), C.byref(self._v(Rc)), (C.c_int64 * 3)(*nf), lo, hi)

    def prolong(self, ec, out, n, lo, hi, correct):
        orc.lib().orc_slab_prolong(C.byref(self._v(ec)), C.byref(self._v(out)), (C.c_int64 * 3)(*n), lo, hi,
                                   int(correct))

    def sync(self):
        pass
