"""ctypes binding of libmg_hip.so (include/mg_hip.h).

There is no CPU fallback: if the shared library is missing or no MI355X is
visible, calls raise.  numpy arrays in, numpy arrays out; all arrays are fp64,
C-contiguous, indexed [i][j][k] like the reference's Field_<double>.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmg_hip.so")

MG_MAX_LEVELS = 12
MG_ABI_VERSION = 1
MG_FLAG_KEEP_FINE_RESIDUAL = 1
MG_FLAG_RESTRICT_7PT = 2

TIMER_KINDS = ("smooth_fine", "residual", "restrict", "smooth_coarse", "prolong",
               "prolong_correct", "norm", "ef")

_i64x3 = C.c_int64 * 3
_f64x3 = C.c_double * 3
_dp = C.POINTER(C.c_double)


class MgConfig(C.Structure):
    _fields_ = [("n", C.c_int64 * 3), ("dh", C.c_double * 3), ("depth", C.c_int32),
                ("nu", C.c_int32 * MG_MAX_LEVELS), ("omega", C.c_double),
                ("device", C.c_int32), ("flags", C.c_int32)]


class MgTimers(C.Structure):
    _fields_ = [("ms", C.c_double * len(TIMER_KINDS)),
                ("launches", C.c_uint64 * len(TIMER_KINDS)),
                ("bytes", C.c_double * len(TIMER_KINDS))]


class MgDevArray(C.Structure):
    """mg_dev_array: planes [i0, i0+planes) of a level's array in device memory."""
    _fields_ = [("ptr", C.c_void_p), ("planes", C.c_int64), ("nj", C.c_int64), ("nk", C.c_int64),
                ("i0", C.c_int64)]


class MgError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libmg_hip error {code}: {text}")
        self.code = code


# every symbol include/mg_hip.h declares: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "mg_abi_version": (C.c_int, []),
    "mg_last_error": (C.c_char_p, []),
    "mg_device_count": (C.c_int, []),
    "mg_default_config": (C.c_int, [C.POINTER(MgConfig), C.c_int64, C.c_int64, C.c_int64, _f64x3, C.c_int]),
    "mg_default_config_b": (C.c_int, [C.POINTER(MgConfig), C.c_int64, C.c_int64, C.c_int64, _f64x3, C.c_int]),
    "mg_create": (C.c_int, [C.POINTER(MgConfig), C.POINTER(_H)]),
    "mg_destroy": (C.c_int, [_H]),
    "mg_set_rho": (C.c_int, [_H, _dp]),
    "mg_set_phi": (C.c_int, [_H, _dp]),
    "mg_get_phi": (C.c_int, [_H, _dp]),
    "mg_solve": (C.c_int, [_H, C.c_uint, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_uint), _dp]),
    "mg_vcycle": (C.c_int, [_H, _dp]),
    "mg_vcycles": (C.c_int, [_H, C.c_uint, _dp]),
    "mg_gs_solve": (C.c_int, [_H, C.c_uint, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_uint), _dp]),
    "mg_compute_ef": (C.c_int, [_H]),
    "mg_get_ef": (C.c_int, [_H, _dp]),
    "mg_host_register": (C.c_int, [C.c_void_p, C.c_uint64]),
    "mg_host_unregister": (C.c_int, [C.c_void_p]),
    "mg_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint64]),
    "mg_host_free": (C.c_int, [C.c_void_p]),
    "mg_reset_state": (C.c_int, [_H]),
    "mg_get_level_array": (C.c_int, [_H, C.c_int, C.c_int, _dp, _i64x3]),
    "mg_set_profiling": (C.c_int, [_H, C.c_int]),
    "mg_get_timers": (C.c_int, [_H, C.POINTER(MgTimers)]),
    "mg_reset_timers": (C.c_int, [_H]),
    "mg_last_call_ms": (C.c_int, [_H, _dp]),
    "mg_op_residual": (C.c_int, [C.c_int, _i64x3, _f64x3, _dp, _dp, _dp, _dp]),
    "mg_op_residual_restrict": (C.c_int, [C.c_int, _i64x3, _f64x3, _dp, _dp, _dp, _dp, _dp]),
    "mg_op_restrict27": (C.c_int, [C.c_int, _i64x3, _dp, _dp]),
    "mg_op_restrict7": (C.c_int, [C.c_int, _i64x3, _dp, _dp]),
    "mg_op_prolong": (C.c_int, [C.c_int, _i64x3, _i64x3, _dp, _dp]),
    "mg_op_correct": (C.c_int, [C.c_int, C.c_int64, _dp, _dp]),
    "mg_op_prolong_correct": (C.c_int, [C.c_int, _i64x3, _dp, _dp]),
    "mg_op_smooth_rb": (C.c_int, [C.c_int, _i64x3, _i64x3, _i64x3, _f64x3, C.c_double, C.c_int, C.c_int, _dp, _dp]),
    "mg_op_compute_ef": (C.c_int, [C.c_int, _i64x3, _f64x3, _dp, _dp]),
    "mg_op_sumsq": (C.c_int, [C.c_int, C.c_int64, _dp, _dp]),
    # device-level operators of the slab-partitioned solve (multigrid_amd/dist.py)
    "mg_dev_pitch": (C.c_int64, [C.c_int64]),
    "mg_level_coef": (C.c_int, [_f64x3, C.c_int, C.c_double * 4]),
    "mg_dev_div_eps0": (C.c_int, [C.c_void_p, C.POINTER(MgDevArray)]),
    "mg_dev_sweep_out_of_place": (C.c_int, [C.POINTER(MgDevArray)]),
    "mg_dev_smooth_rb": (C.c_int, [C.c_void_p, C.POINTER(MgDevArray), C.POINTER(MgDevArray), C.POINTER(MgDevArray),
                                   _i64x3, C.c_double * 4, C.c_double, C.c_int, C.c_int64, C.c_int64,
                                   C.c_int64, C.c_int64, C.POINTER(C.c_int)]),
    "mg_dev_residual_restrict": (C.c_int, [C.c_void_p, C.POINTER(MgDevArray), C.POINTER(MgDevArray),
                                           C.POINTER(MgDevArray), _i64x3, C.c_double * 4, C.c_void_p, C.c_int64,
                                           C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "mg_dev_residual_restrict_noalloc": (C.c_int, [C.c_void_p, C.POINTER(MgDevArray), C.POINTER(MgDevArray),
                                                   C.POINTER(MgDevArray), _i64x3, C.c_double * 4, C.c_int64,
                                                   C.c_int64, C.c_void_p]),
    "mg_dev_sum_partials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mg_dev_restrict27": (C.c_int, [C.c_void_p, C.POINTER(MgDevArray), C.POINTER(MgDevArray), _i64x3,
                                    C.c_int64, C.c_int64]),
    "mg_dev_prolong": (C.c_int, [C.c_void_p, C.POINTER(MgDevArray), C.POINTER(MgDevArray), _i64x3, C.c_int64,
                                 C.c_int64, C.c_int]),
    "mg_dev_prolong_smooth_rb": (C.c_int, [C.c_void_p, C.POINTER(MgDevArray), C.c_int, C.POINTER(MgDevArray),
                                           C.POINTER(MgDevArray), C.POINTER(MgDevArray), _i64x3, C.c_double * 4,
                                           C.c_double, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                           C.POINTER(C.c_int)]),
    "mg_dev_onload_folds": (C.c_int, [C.POINTER(MgDevArray), C.c_int]),
    "mg_dev_compute_ef": (C.c_int, [C.c_void_p, C.POINTER(MgDevArray), C.c_void_p, C.c_int64, _i64x3, _f64x3,
                                    C.c_int64, C.c_int64]),
}

_lib = None


def load():
    """Load libmg_hip.so and bind every declared symbol; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MgError(-100, f"{LIB_PATH} not built (run python -m multigrid_amd.build)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(lib, name)  # AttributeError if the export is missing
            f.restype = res
            f.argtypes = args
        if lib.mg_abi_version() != MG_ABI_VERSION:
            raise MgError(-101, "ABI version mismatch between libmg_hip.so and libmg.py")
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise MgError(rc, load().mg_last_error().decode(errors="replace"))


def _in(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _out(a):
    if a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"] or not a.flags["WRITEABLE"]:
        raise ValueError("output array must be writable C-contiguous float64")
    return a.ctypes.data_as(_dp)


def device_count():
    return load().mg_device_count()


def default_config(n, dh, depth, device=0, variant_b=False):
    cfg = MgConfig()
    f = load().mg_default_config_b if variant_b else load().mg_default_config
    _check(f(C.byref(cfg), n[0], n[1], n[2], _f64x3(*dh), depth))
    cfg.device = device
    return cfg


class Handle:
    """Owns one mg_handle (one reference MultiGridSolverVn object's worth of state)."""

    def __init__(self, n, dh, depth, nu=None, omega=None, device=0, flags=0, variant_b=False):
        self.n = tuple(int(a) for a in n)
        self.depth = depth
        cfg = default_config(self.n, dh, depth, device, variant_b)
        if nu is not None:
            if len(nu) != depth + 1:
                raise ValueError("nu must have depth+1 entries")
            for l, v in enumerate(nu):
                cfg.nu[l] = v
        if omega is not None:
            cfg.omega = omega
        cfg.flags |= flags
        self.cfg = cfg
        self._h = _H()
        _check(load().mg_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            load().mg_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @staticmethod
    def _shape(a, want, what):
        """The C library copies prod(want) doubles: a wrong shape must never reach it."""
        if tuple(a.shape) != tuple(want):
            raise ValueError(f"{what} has shape {tuple(a.shape)}, expected {tuple(want)}")

    def set_rho(self, rho):
        a, p = _in(rho)
        self._shape(a, self.n, "rho")
        _check(load().mg_set_rho(self._h, p))

    def set_phi(self, phi):
        a, p = _in(phi)
        self._shape(a, self.n, "phi")
        _check(load().mg_set_phi(self._h, p))

    def get_phi(self, out=None):
        if out is None:
            out = np.empty(self.n)
        self._shape(out, self.n, "out")
        _check(load().mg_get_phi(self._h, _out(out)))
        return out

    def solve(self, max_it=10000, tol=1e-4):
        conv, cyc, l2 = C.c_int(0), C.c_uint(0), C.c_double(0)
        _check(load().mg_solve(self._h, max_it, tol, C.byref(conv), C.byref(cyc), C.byref(l2)))
        return bool(conv.value), cyc.value, l2.value

    def vcycles(self, count=1):
        l2 = C.c_double(0)
        _check(load().mg_vcycles(self._h, count, C.byref(l2)))
        return l2.value

    def gs_solve(self, max_it=10000, tol=1e-4):
        """Plain red-black GS-SOR (the reference's GaussSeidelSolver loop). Returns (converged, sweeps, L2)."""
        conv, its, l2 = C.c_int(0), C.c_uint(0), C.c_double(0)
        _check(load().mg_gs_solve(self._h, max_it, tol, C.byref(conv), C.byref(its), C.byref(l2)))
        return bool(conv.value), its.value, l2.value

    def compute_ef(self, download=True):
        """E = -grad(phi) on the device; returns the (ni,nj,nk,3) array unless download is False."""
        _check(load().mg_compute_ef(self._h))
        return self.get_ef() if download else None

    def get_ef(self):
        out = np.empty(self.n + (3,))
        _check(load().mg_get_ef(self._h, _out(out)))
        return out

    def reset_state(self):
        _check(load().mg_reset_state(self._h))

    def level_array(self, which, level):
        """which: 'R' or 'eps'."""
        w = {"R": 0, "eps": 1}[which]
        shp = tuple(a >> level for a in self.n)
        if w == 1:
            shp = tuple(a + 1 for a in shp)
        out = np.empty(shp)
        dims = _i64x3()
        _check(load().mg_get_level_array(self._h, w, level, _out(out), dims))
        if tuple(dims) != shp:
            raise MgError(-102, f"level array dims {tuple(dims)} differ from the expected {shp}")
        return out

    def set_profiling(self, on=True):
        _check(load().mg_set_profiling(self._h, int(on)))

    def reset_timers(self):
        _check(load().mg_reset_timers(self._h))

    def timers(self):
        t = MgTimers()
        _check(load().mg_get_timers(self._h, C.byref(t)))
        return {k: {"ms": t.ms[i], "launches": int(t.launches[i]), "bytes": t.bytes[i]}
                for i, k in enumerate(TIMER_KINDS)}

    def last_call_ms(self):
        ms = C.c_double(0)
        _check(load().mg_last_call_ms(self._h, C.byref(ms)))
        return ms.value


# ---- single operators ------------------------------------------------------

def op_residual(phi, rho, dh, device=0):
    phi, pp = _in(phi)
    rho, pr = _in(rho)
    R = np.empty_like(phi)
    s = C.c_double(0)
    _check(load().mg_op_residual(device, _i64x3(*phi.shape), _f64x3(*dh), pp, pr, _out(R), C.byref(s)))
    return R, s.value


def op_residual_restrict(phi, rho, dh, Rc=None, want_fine=False, device=0):
    """Returns (R_fine or None, R_coarse, sum of squares)."""
    phi, pp = _in(phi)
    rho, pr = _in(rho)
    if Rc is None:
        Rc = np.zeros(tuple(s >> 1 for s in phi.shape))
    Rf = np.empty_like(phi) if want_fine else None
    s = C.c_double(0)
    _check(load().mg_op_residual_restrict(device, _i64x3(*phi.shape), _f64x3(*dh), pp, pr,
                                          _out(Rf) if want_fine else None, _out(Rc), C.byref(s)))
    return Rf, Rc, s.value


def op_restrict27(Rf, Rc=None, device=0):
    Rf, pf = _in(Rf)
    if Rc is None:
        Rc = np.zeros(tuple(s >> 1 for s in Rf.shape))
    _check(load().mg_op_restrict27(device, _i64x3(*Rf.shape), pf, _out(Rc)))
    return Rc


def op_restrict7(Rf, Rc=None, device=0):
    Rf, pf = _in(Rf)
    if Rc is None:
        Rc = np.zeros(tuple(s >> 1 for s in Rf.shape))
    _check(load().mg_op_restrict7(device, _i64x3(*Rf.shape), pf, _out(Rc)))
    return Rc


def op_prolong(ec, ef, n, device=0):
    ec, pc = _in(ec)
    Handle._shape(ec, tuple((a >> 1) + 1 for a in n), "ec")
    if len(ef.shape) != 3 or any(f < a for f, a in zip(ef.shape, n)):
        raise ValueError(f"ef has shape {tuple(ef.shape)}, needs at least {tuple(n)}")
    _check(load().mg_op_prolong(device, _i64x3(*n), _i64x3(*ef.shape), pc, _out(ef)))
    return ef


def op_correct(phi, eps, device=0):
    eps, pe = _in(eps)
    _check(load().mg_op_correct(device, phi.size, _out(phi), pe))
    return phi


def op_prolong_correct(ec, phi, device=0):
    ec, pc = _in(ec)
    Handle._shape(ec, tuple((a >> 1) + 1 for a in phi.shape), "ec")
    _check(load().mg_op_prolong_correct(device, _i64x3(*phi.shape), pc, _out(phi)))
    return phi


def op_smooth_rb(u, rhs, n, c, omega=1.4, sweeps=1, rhs_is_rho=False, device=0):
    rhs, pr = _in(rhs)
    _check(load().mg_op_smooth_rb(device, _i64x3(*n), _i64x3(*u.shape), _i64x3(*rhs.shape),
                                  _f64x3(*c[:3]), omega, sweeps, int(rhs_is_rho), pr, _out(u)))
    return u


def op_compute_ef(phi, dh, device=0):
    phi, pp = _in(phi)
    ef = np.empty(phi.shape + (3,))
    _check(load().mg_op_compute_ef(device, _i64x3(*phi.shape), _f64x3(*dh), pp, _out(ef)))
    return ef


def op_sumsq(x, device=0):
    x, px = _in(x)
    s = C.c_double(0)
    _check(load().mg_op_sumsq(device, x.size, px, C.byref(s)))
    return s.value
