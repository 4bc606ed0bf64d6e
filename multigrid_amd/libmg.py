"""ctypes binding of libmg_hip.so (include/mg_hip.h).

There is no CPU fallback: if the shared library is missing or no MI355X is
visible, calls raise.  numpy arrays in, numpy arrays out; all arrays are fp64,
C-contiguous, indexed [i][j][k] like the reference's Field_<double>.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MG_HIP_LIB names another build of the library (used to time two builds side by side)
LIB_PATH = os.environ.get("MG_HIP_LIB") or os.path.join(_HERE, "libmg_hip.so")

MG_MAX_LEVELS = 12
MG_ABI_VERSION = 2
MG_MAX_GPUS = 16
MG_UNIQUE_ID_BYTES = 128
MG_TRANSPORT_RCCL, MG_TRANSPORT_LOCAL, MG_TRANSPORT_NONE = 0, 1, 2
MG_FLAG_KEEP_FINE_RESIDUAL = 1
MG_FLAG_RESTRICT_7PT = 2
MG_FLAG_RESERVE_EF = 4
MG_FLAG_FAST_SMOOTHER = 8

TIMER_KINDS = ("smooth_fine", "residual", "restrict", "smooth_coarse", "prolong",
               "prolong_correct", "norm", "ef")

_i64x3 = C.c_int64 * 3
_f64x3 = C.c_double * 3
_dp = C.POINTER(C.c_double)


class MgConfig(C.Structure):
    _fields_ = [("n", C.c_int64 * 3), ("dh", C.c_double * 3), ("depth", C.c_int32),
                ("nu", C.c_int32 * MG_MAX_LEVELS), ("omega", C.c_double),
                ("device", C.c_int32), ("flags", C.c_int32), ("n_gpus", C.c_int32),
                ("devices", C.c_int32 * MG_MAX_GPUS), ("coarse_cut", C.c_int32)]


class MgDistConfig(C.Structure):
    _fields_ = [("base", MgConfig), ("rank", C.c_int32), ("world", C.c_int32), ("coarse_cut", C.c_int32),
                ("transport", C.c_int32), ("no_overlap", C.c_int32), ("unique_id", C.c_uint8 * MG_UNIQUE_ID_BYTES),
                ("local_group", C.c_void_p)]


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
    "mg_set_exact_division": (C.c_int, [C.c_int]),
    "mg_default_config": (C.c_int, [C.POINTER(MgConfig), C.c_int64, C.c_int64, C.c_int64, _f64x3, C.c_int]),
    "mg_default_config_b": (C.c_int, [C.POINTER(MgConfig), C.c_int64, C.c_int64, C.c_int64, _f64x3, C.c_int]),
    "mg_create": (C.c_int, [C.POINTER(MgConfig), C.POINTER(_H)]),
    "mg_device_bytes": (C.c_int, [C.POINTER(MgConfig), C.POINTER(C.c_uint64)]),
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
    "mg_dev_sweeps_in_pairs": (C.c_int, [C.POINTER(MgDevArray), C.POINTER(MgDevArray), _i64x3, C.c_int64, C.c_int64]),
    "mg_dev_compute_ef": (C.c_int, [C.c_void_p, C.POINTER(MgDevArray), C.c_void_p, C.c_int64, _i64x3, _f64x3,
                                    C.c_int64, C.c_int64]),
    # slab-partitioned solve driven inside the library, one rank per GPU
    "mg_dist_unique_id": (C.c_int, [C.c_uint8 * MG_UNIQUE_ID_BYTES]),
    "mg_local_group_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mg_local_group_destroy": (C.c_int, [C.c_void_p]),
    "mg_dist_create": (C.c_int, [C.POINTER(MgDistConfig), C.POINTER(C.c_void_p)]),
    "mg_dist_destroy": (C.c_int, [C.c_void_p]),
    "mg_dist_plan": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int64 * 2, C.c_int64 * 2, C.c_int64 * 2]),
    "mg_dist_set_rho_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp]),
    "mg_dist_set_phi_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp]),
    "mg_dist_get_phi_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp]),
    "mg_dist_vcycles": (C.c_int, [C.c_void_p, C.c_uint, _dp]),
    "mg_dist_solve": (C.c_int, [C.c_void_p, C.c_uint, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_uint), _dp]),
    "mg_dist_compute_ef": (C.c_int, [C.c_void_p]),
    "mg_dist_get_ef_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp]),
    "mg_dist_sync": (C.c_int, [C.c_void_p]),
    "mg_dist_abort": (C.c_int, [C.c_void_p]),
    "mg_dist_stats": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_uint64)]),
    "mg_dist_deep_halos": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mg_dist_comm_log": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64)]),
    "mg_dist_exchange_plan": (C.c_int, [C.POINTER(MgDistConfig), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.c_int64 * 2, C.c_int64 * 2, C.c_int64 * 2, C.c_int32 * 4]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm ships its own copy of the HIP runtime (same soname as the system's).  A process
    that ends up with both copies initialised finds no device in the second one.  If torch is
    installed but not imported yet, map ITS copy first: libmg_hip.so then binds to it, and a later
    `import torch` finds its runtime already loaded -- one runtime whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch's runtime is already mapped
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations if spec and spec.submodule_search_locations else []):
        path = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass  # fall back to the system runtime
            return


def load():
    """Load libmg_hip.so and bind every declared symbol; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MgError(-100, f"{LIB_PATH} not built (run python -m multigrid_amd.build)")
        _share_hip_runtime_with_torch()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(lib, name)  # AttributeError if the export is missing
            f.restype = res
            f.argtypes = args
        if lib.mg_abi_version() != MG_ABI_VERSION:
            raise MgError(-101, "ABI version mismatch between libmg_hip.so and libmg.py")
        _lib = lib
    return _lib


def device_bytes(n, dh, depth, flags=0, devices=None, coarse_cut=0):
    """Device memory in bytes a Handle of these arguments allocates (per GPU: the largest share);
    arithmetic only, no GPU needed."""
    cfg = default_config(tuple(int(a) for a in n), dh, depth, 0)
    cfg.flags |= flags
    if devices is not None and len(devices) > 1:
        cfg.n_gpus = len(devices)
        cfg.coarse_cut = coarse_cut
    out = C.c_uint64(0)
    _check(load().mg_device_bytes(C.byref(cfg), C.byref(out)))
    return int(out.value)


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

    def __init__(self, n, dh, depth, nu=None, omega=None, device=0, flags=0, variant_b=False, devices=None,
                 coarse_cut=0):
        """devices: a list of two or more GPU ordinals splits the grid into slabs over them (one host
        thread per GPU inside the library); an ordinal may repeat (several slabs on one GPU)."""
        self.n = tuple(int(a) for a in n)
        self.depth = depth
        cfg = default_config(self.n, dh, depth, device, variant_b)
        if devices is not None and len(devices) > 1:
            if len(devices) > MG_MAX_GPUS:
                raise ValueError(f"at most {MG_MAX_GPUS} devices")
            cfg.n_gpus = len(devices)
            for i, d in enumerate(devices):
                cfg.devices[i] = d
            cfg.coarse_cut = coarse_cut
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


# ---- slab-partitioned solve, one rank per GPU (mg_dist_*) --------------------

def dist_unique_id():
    """128 bytes from rank 0 that every rank of an RCCL group must be created with."""
    buf = (C.c_uint8 * MG_UNIQUE_ID_BYTES)()
    _check(load().mg_dist_unique_id(buf))
    return bytes(buf)


class LocalGroup:
    """Meeting point of ranks that are threads of this process (MG_TRANSPORT_LOCAL)."""

    def __init__(self, world, hold_back=False):
        self._g = C.c_void_p()
        _check(load().mg_local_group_create(world, int(hold_back), C.byref(self._g)))

    def close(self):
        if getattr(self, "_g", None):
            load().mg_local_group_destroy(self._g)
            self._g = None

    __del__ = close


def dist_exchange_plan(n, depth, rank, world, level, coarse_cut=0, nu=None, deep_halos=True, variant_b=False):
    """mg_dist_exchange_plan: what rank `rank` of `world` exchanges on `level` once per cycle, from
    the configuration alone (no device).  Returns a dict with deep, cut, own, first, last and
    halo = (received below, sent from the low end, sent from the high end, received above)."""
    n = tuple(int(a) for a in n)
    dc = MgDistConfig()
    dc.base = default_config(n, (1.0, 1.0, 1.0), depth, 0, variant_b=variant_b) if variant_b else \
        default_config(n, (1.0, 1.0, 1.0), depth, 0)
    if nu is not None:
        for l, v in enumerate(nu):
            dc.base.nu[l] = v
    dc.rank, dc.world, dc.coarse_cut, dc.no_overlap = rank, world, coarse_cut, 0 if deep_halos else 2
    deep, cut = C.c_int32(0), C.c_int32(0)
    own, first, last, halo = (C.c_int64 * 2)(), (C.c_int64 * 2)(), (C.c_int64 * 2)(), (C.c_int32 * 4)()
    _check(load().mg_dist_exchange_plan(C.byref(dc), level, C.byref(deep), C.byref(cut), own, first, last, halo))
    return {"deep": bool(deep.value), "cut": cut.value, "own": tuple(own), "first": tuple(first),
            "last": tuple(last), "halo": tuple(halo)}


class DistRank:
    """One rank of a slab-partitioned solve (mg_dist): construct on every rank of the group with the
    same arguments apart from rank and device; creation, vcycles, solve and close are collective."""

    def __init__(self, n, dh, depth, rank, world, device=0, unique_id=None, local_group=None, coarse_cut=0,
                 nu=None, omega=None, overlap=True, alone=False, flags=0, deep_halos=True):
        self.n = tuple(int(a) for a in n)
        dc = MgDistConfig()
        dc.base = default_config(self.n, dh, depth, device)
        dc.base.flags |= flags
        if nu is not None:
            for l, v in enumerate(nu):
                dc.base.nu[l] = v
        if omega is not None:
            dc.base.omega = omega
        # no_overlap: bit 0 = halo exchanges after whole sweeps, bit 1 = an exchange after every sweep
        # on every level (no deep halos)
        dc.rank, dc.world, dc.coarse_cut, dc.no_overlap = rank, world, coarse_cut, int(not overlap) | (0 if deep_halos else 2)
        if alone:
            dc.transport = MG_TRANSPORT_NONE  # this rank of `world`, without peers: timing only
        elif local_group is not None or (world == 1 and unique_id is None):
            dc.transport = MG_TRANSPORT_LOCAL
            dc.local_group = local_group._g if local_group is not None else None
        else:
            if unique_id is None or len(unique_id) != MG_UNIQUE_ID_BYTES:
                raise ValueError("an RCCL rank needs the 128-byte unique id made on rank 0")
            dc.transport = MG_TRANSPORT_RCCL
            C.memmove(dc.unique_id, unique_id, MG_UNIQUE_ID_BYTES)
        self._d = C.c_void_p()
        _check(load().mg_dist_create(C.byref(dc), C.byref(self._d)))
        cut = C.c_int32(0)
        own, sp, sr = (C.c_int64 * 2)(), (C.c_int64 * 2)(), (C.c_int64 * 2)()
        _check(load().mg_dist_plan(self._d, C.byref(cut), own, sp, sr))
        self.cut, self.own, self.stored_phi, self.stored_rho = cut.value, tuple(own), tuple(sp), tuple(sr)
        deep = C.c_int32(0)
        _check(load().mg_dist_deep_halos(self._d, C.byref(deep)))
        self.deep_halos = bool(deep.value)

    def close(self):
        if getattr(self, "_d", None):
            load().mg_dist_destroy(self._d)
            self._d = None

    def comm_log(self):
        """mg_dist_comm_log: the transfers a rank created with alone=True would have made, as a list of
        (what, peer, array, lo, hi)."""
        count = C.c_int64(0)
        _check(load().mg_dist_comm_log(self._d, None, 0, C.byref(count)))
        buf = (C.c_int64 * (5 * max(count.value, 1)))()
        _check(load().mg_dist_comm_log(self._d, buf, count.value, C.byref(count)))
        return [tuple(buf[5 * e:5 * e + 5]) for e in range(count.value)]

    def _planes(self, f, rng, of_planes, chunk):
        lo, hi = rng
        for a in range(lo, hi, chunk):
            z = min(a + chunk, hi)
            arr, p = _in(of_planes(a, z))
            Handle._shape(arr, (z - a,) + self.n[1:], "planes")
            _check(f(self._d, a, z, p))

    def set_rho(self, rho_of_planes, chunk=32):
        """rho_of_planes(lo, hi) -> dense array of global planes [lo,hi); called for a few planes at a
        time, so a rank never holds its whole slab in host memory."""
        self._planes(load().mg_dist_set_rho_planes, self.stored_rho, rho_of_planes, chunk)

    def set_phi(self, phi_of_planes, chunk=32):
        self._planes(load().mg_dist_set_phi_planes, self.stored_phi, phi_of_planes, chunk)

    def get_phi_owned(self):
        lo, hi = self.own
        out = np.empty((hi - lo,) + self.n[1:])
        _check(load().mg_dist_get_phi_planes(self._d, lo, hi, _out(out)))
        return lo, out

    def vcycles(self, count=1):
        l2 = C.c_double(0)
        _check(load().mg_dist_vcycles(self._d, count, C.byref(l2)))
        return l2.value

    def solve(self, max_it=10000, tol=1e-4):
        conv, cyc, l2 = C.c_int(0), C.c_uint(0), C.c_double(0)
        _check(load().mg_dist_solve(self._d, max_it, tol, C.byref(conv), C.byref(cyc), C.byref(l2)))
        return bool(conv.value), cyc.value, l2.value

    def compute_ef_owned(self):
        """(first owned plane, E on the owned planes as (planes, nj, nk, 3)); collective."""
        _check(load().mg_dist_compute_ef(self._d))
        lo, hi = self.own
        out = np.empty((hi - lo,) + self.n[1:] + (3,))
        _check(load().mg_dist_get_ef_planes(self._d, lo, hi, _out(out)))
        return lo, out

    def sync(self):
        _check(load().mg_dist_sync(self._d))

    def abort(self):
        """Give up this rank's communication (callable from any thread): see mg_dist_abort."""
        _check(load().mg_dist_abort(self._d))

    def stats(self):
        ms, sent = C.c_double(0), C.c_uint64(0)
        _check(load().mg_dist_stats(self._d, C.byref(ms), C.byref(sent)))
        return {"last_ms": ms.value, "bytes_sent": int(sent.value)}
