"""ctypes/numpy front end of oracle/mg_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It is the checker for the HIP library, never a fallback for it.
See the header of mg_oracle.c for what is restated, the reference lines each
function follows and the parity status ("parity unpinned" by reference-held
fixtures; cross-checked against survey-recorded residual histories).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmg_oracle.so")

EPS_0 = 8.85418782e-12  # cpp/World.h:14

_i64x3 = C.c_int64 * 3
_f64x3 = C.c_double * 3
_f64x4 = C.c_double * 4
_dp = C.POINTER(C.c_double)


def build(force=False):
    """Compile libmg_oracle.so with the committed Makefile (gcc -O3, contraction off)."""
    src = os.path.join(_HERE, "mg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmg_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class OrcView(C.Structure):
    """orc_view: planes [i0, i0+d[0]) of a level's array, row pitch sj."""
    _fields_ = [("p", _dp), ("d", C.c_int64 * 3), ("si", C.c_int64), ("sj", C.c_int64), ("i0", C.c_int64)]


def view(a, nk, i0):
    """orc_view of a C-contiguous float64 array a[planes][nj][pitch] holding rows of nk nodes."""
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and a.ndim == 3
    v = OrcView()
    v.p = a.ctypes.data_as(_dp)
    v.d[0], v.d[1], v.d[2] = a.shape[0], a.shape[1], nk
    v.sj = a.shape[2]
    v.si = a.shape[1] * a.shape[2]
    v.i0 = i0
    return v


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_coeffs.argtypes = [_f64x3, C.c_int, _f64x4]
        L.orc_coeffs.restype = None
        sm = [_dp, _i64x3, _dp, _i64x3, _i64x3, _f64x4, C.c_double, C.c_int, C.c_int]
        L.orc_smooth_lex.argtypes = sm
        L.orc_smooth_lex.restype = None
        L.orc_smooth_rb.argtypes = sm
        L.orc_smooth_rb.restype = None
        L.orc_smooth_rb_fast.argtypes = sm
        L.orc_smooth_rb_fast.restype = None
        L.orc_residual.argtypes = [_dp, _dp, _dp, _i64x3, _f64x4]
        L.orc_residual.restype = C.c_double
        L.orc_sumsq.argtypes = [_dp, C.c_int64]
        L.orc_sumsq.restype = C.c_double
        L.orc_restrict27.argtypes = [_dp, _i64x3, _dp, _i64x3]
        L.orc_restrict27.restype = None
        L.orc_restrict7.argtypes = [_dp, _i64x3, _dp, _i64x3]
        L.orc_restrict7.restype = None
        L.orc_solver_set_restrict7.argtypes = [C.c_void_p, C.c_int]
        L.orc_solver_set_restrict7.restype = None
        L.orc_prolong.argtypes = [_dp, _i64x3, _dp, _i64x3, _i64x3]
        L.orc_prolong.restype = None
        L.orc_correct.argtypes = [_dp, _dp, C.c_int64]
        L.orc_correct.restype = None
        L.orc_compute_ef.argtypes = [_dp, _dp, _i64x3, _f64x3]
        L.orc_compute_ef.restype = None
        L.orc_solver_new.argtypes = [_i64x3, _f64x3, C.c_int, C.POINTER(C.c_int), C.c_double]
        L.orc_solver_new.restype = C.c_void_p
        L.orc_solver_free.argtypes = [C.c_void_p]
        L.orc_solver_free.restype = None
        L.orc_solver_reset.argtypes = [C.c_void_p]
        L.orc_solver_reset.restype = None
        L.orc_solver_R.argtypes = [C.c_void_p, C.c_int]
        L.orc_solver_R.restype = _dp
        L.orc_solver_eps.argtypes = [C.c_void_p, C.c_int]
        L.orc_solver_eps.restype = _dp
        L.orc_solve.argtypes = [C.c_void_p, _dp, _dp, C.c_uint, C.c_double, C.c_int,
                                C.POINTER(C.c_uint), _dp, _dp]
        L.orc_solve.restype = C.c_int
        L.orc_gs_solve.argtypes = [_dp, _dp, _i64x3, _f64x3, C.c_uint, C.c_double, C.c_double, C.c_int,
                                   C.POINTER(C.c_uint), _dp]
        L.orc_gs_solve.restype = C.c_int
        V = C.POINTER(OrcView)
        L.orc_slab_smooth_rb.argtypes = [V, V, _i64x3, _f64x4, C.c_double, C.c_int64, C.c_int64]
        L.orc_slab_smooth_rb.restype = None
        L.orc_slab_residual_restrict.argtypes = [V, V, V, _i64x3, _f64x4, C.c_int64, C.c_int64]
        L.orc_slab_residual_restrict.restype = C.c_double
        L.orc_slab_restrict27.argtypes = [V, V, _i64x3, C.c_int64, C.c_int64]
        L.orc_slab_restrict27.restype = None
        L.orc_slab_prolong.argtypes = [V, V, _i64x3, C.c_int64, C.c_int64, C.c_int]
        L.orc_slab_prolong.restype = None
        L.orc_slab_div_eps0.argtypes = [V]
        L.orc_slab_div_eps0.restype = None
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def _dims(a):
    return _i64x3(*a.shape)


def coeffs(dh, level):
    """(cx, cy, cz, d) of level `level` for spacing dh[3]."""
    out = _f64x4()
    lib().orc_coeffs(_f64x3(*dh), level, out)
    return tuple(out)


def level_sizes(n0, depth):
    """n_l = n_0 >> l for l = 0..depth (per axis)."""
    return [tuple(int(a) >> l for a in n0) for l in range(depth + 1)]


def smooth(u, rhs, n, c, omega=1.4, sweeps=1, div_eps0=False, rb=False, fast=False):
    """In-place GS-SOR sweeps on u (lexicographic, or red-black when rb; fast: the red-black sweep
    with the point update in its short form, the HIP library's MG_FLAG_FAST_SMOOTHER)."""
    f = lib().orc_smooth_rb_fast if fast else lib().orc_smooth_rb if rb else lib().orc_smooth_lex
    f(_p(u), _dims(u), _p(rhs), _dims(rhs), _i64x3(*n), _f64x4(*c), omega, sweeps, int(div_eps0))
    return u


def residual(phi, rho, c):
    """Returns (R, sum of R^2 in i,j,k order)."""
    R = np.empty_like(phi)
    s = lib().orc_residual(_p(phi), _p(rho), _p(R), _dims(phi), _f64x4(*c))
    return R, s


def sumsq(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().orc_sumsq(_p(x), x.size)


def restrict27(Rf, Rc=None):
    """Full-weighting restriction; Rc (dims nf>>1) is created zero-filled if not given."""
    if Rc is None:
        Rc = np.zeros(tuple(s >> 1 for s in Rf.shape))
    lib().orc_restrict27(_p(Rf), _dims(Rf), _p(Rc), _dims(Rc))
    return Rc


def restrict7(Rf, Rc=None):
    """The "B" classes' 7-point restriction; Rc (dims nf>>1) is created zero-filled if not given."""
    if Rc is None:
        Rc = np.zeros(tuple(s >> 1 for s in Rf.shape))
    lib().orc_restrict7(_p(Rf), _dims(Rf), _p(Rc), _dims(Rc))
    return Rc


def prolong(ec, ef, n):
    """Overwrite ef on 0..n-2 per axis with the trilinear interpolant of ec."""
    lib().orc_prolong(_p(ec), _dims(ec), _p(ef), _dims(ef), _i64x3(*n))
    return ef


def correct(phi, eps):
    lib().orc_correct(_p(phi), _p(eps), phi.size)
    return phi


def compute_ef(phi, dh):
    ef = np.empty(phi.shape + (3,))
    lib().orc_compute_ef(_p(phi), _p(ef), _dims(phi), _f64x3(*dh))
    return ef


def gs_solve(phi, rho, dh, max_it=10000, tol=1e-4, omega=1.4, rb=False):
    """The reference's GaussSeidelSolver loop in place on phi. Returns (converged, sweeps, L2)."""
    its = C.c_uint(0)
    l2 = C.c_double(0)
    ok = lib().orc_gs_solve(_p(phi), _p(rho), _dims(phi), _f64x3(*dh), max_it, tol, omega, int(rb),
                            C.byref(its), C.byref(l2))
    return bool(ok), its.value, l2.value


SCHEDULES = {  # nu[0..L] of the reference classes (cpp/PotentialSolver.cpp:2422-2425,
    1: (3, 50),                 # :2629-2634, :2922-2929, :3298-3307, :3759-3770)
    2: (3, 5, 50),
    3: (3, 5, 5, 50),
    4: (3, 5, 5, 5, 50),
    5: (1, 5, 5, 5, 5, 50),
}


def schedule(depth, variant_b=False):
    """Reference schedule for depth 1..5; beyond that the V5 pattern extended.  variant_b: the
    "B" classes, whose V5B runs 1,2,3,4,5,50 (cpp/PotentialSolver.cpp:1928-1939)."""
    if variant_b and depth >= 5:
        return (1,) + tuple(min(l + 1, 5) for l in range(1, depth)) + (50,)
    if depth in SCHEDULES:
        return SCHEDULES[depth]
    return (1,) + (5,) * (depth - 1) + (50,)


class Solver:
    """The state a reference MultiGridSolver* object keeps between solve() calls."""

    def __init__(self, n0, dh, depth, nu=None, omega=1.4, variant_b=False):
        self.n0 = tuple(int(a) for a in n0)
        self.dh = tuple(float(a) for a in dh)
        self.depth = depth
        self.nu = tuple(nu) if nu is not None else schedule(depth, variant_b)
        assert len(self.nu) == depth + 1
        self.omega = omega
        arr = (C.c_int * (depth + 1))(*self.nu)
        self._h = lib().orc_solver_new(_i64x3(*self.n0), _f64x3(*self.dh), depth, arr, omega)
        if not self._h:
            raise ValueError("oracle: invalid size/depth")
        lib().orc_solver_set_restrict7(self._h, int(variant_b))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_solver_free(self._h)
            self._h = None

    def reset(self):
        lib().orc_solver_reset(self._h)

    def _view(self, ptr, shape):
        return np.ctypeslib.as_array(ptr, shape=shape)

    def R(self, l):
        return self._view(lib().orc_solver_R(self._h, l), tuple(a >> l for a in self.n0))

    def eps(self, l):
        shp = tuple(a >> l for a in self.n0)
        if l > 0:
            shp = tuple(a + 1 for a in shp)
        return self._view(lib().orc_solver_eps(self._h, l), shp)

    def solve(self, phi, rho, max_it=10000, tol=1e-4, rb=False, fast=False):
        """Runs the reference iteration loop in place on phi (rb: red-black smoother; fast: red-black
        with the short-form point update on the levels >= 1).  Returns (converged, cycles, L2, l2_history)."""
        if fast:
            rb = 2
        hist = np.zeros(max_it)
        cyc = C.c_uint(0)
        l2 = C.c_double(0)
        ok = lib().orc_solve(self._h, _p(phi), _p(rho), max_it, tol, int(rb),
                             C.byref(cyc), C.byref(l2), _p(hist))
        return bool(ok), cyc.value, l2.value, hist[:cyc.value if ok else max_it].copy()


def main_extents_dh(n):
    """dh for the reference driver's box, cpp/Main.cpp:37 and cpp/World.cpp:29-30."""
    x0 = (-0.2, -0.2, -0.1)
    xm = (0.2, 0.2, 0.3)
    return tuple((xm[a] - x0[a]) / (n[a] - 1) for a in range(3))


def synthetic_rho(n):
    """Deterministic charge density used for every benchmark and known-answer run:
    rho = EPS_0*1000*(sin(pi x)sin(pi y)sin(pi z) + 0.5 sin(5 pi x)sin(3 pi y)sin(7 pi z)),
    x,y,z = i/(ni-1), j/(nj-1), k/(nk-1)."""
    x = np.arange(n[0]) / (n[0] - 1)
    y = np.arange(n[1]) / (n[1] - 1)
    z = np.arange(n[2]) / (n[2] - 1)
    pi = math.pi
    a = np.sin(pi * x)[:, None, None] * np.sin(pi * y)[None, :, None] * np.sin(pi * z)[None, None, :]
    b = np.sin(5 * pi * x)[:, None, None] * np.sin(3 * pi * y)[None, :, None] * np.sin(7 * pi * z)[None, None, :]
    return np.ascontiguousarray(EPS_0 * 1000.0 * (a + 0.5 * b))
