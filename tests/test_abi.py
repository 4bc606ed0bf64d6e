"""CPU-side checks of the C-ABI library: it loads, exports every symbol the
header declares, and refuses to compute without a GPU instead of falling back."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mg_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_binds(mg):
    assert declared_symbols() == sorted(mg.SYMBOLS)


def test_library_exports_every_declared_symbol(mg):
    lib = C.CDLL(mg.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"libmg_hip.so does not export {name}"


def test_abi_version(mg):
    assert mg.load().mg_abi_version() == mg.MG_ABI_VERSION


def test_default_config_matches_reference_schedules(mg):
    # cpp/PotentialSolver.cpp:2422-2425, :2629-2634, :2922-2929, :3298-3307, :3759-3770
    want = {1: [3, 50], 2: [3, 5, 50], 3: [3, 5, 5, 50], 4: [3, 5, 5, 5, 50], 5: [1, 5, 5, 5, 5, 50]}
    for depth, nu in want.items():
        cfg = mg.default_config((513, 513, 513), (1e-3, 1e-3, 1e-3), depth)
        assert list(cfg.nu[: depth + 1]) == nu
        assert cfg.omega == 1.4
        assert cfg.depth == depth


def test_bad_arguments_are_rejected(mg):
    with pytest.raises(mg.MgError):
        mg.default_config((41, 41, 41), (1e-2,) * 3, -1)
    assert mg.default_config((41, 41, 41), (1e-2,) * 3, 0).depth == 0  # Gauss-Seidel / field-only handle
    with pytest.raises(mg.MgError):
        mg.default_config((41, 41, 41), (1e-2,) * 3, mg.MG_MAX_LEVELS)


def test_device_memory_of_a_configuration_is_plain_arithmetic(mg):
    """mg_device_bytes needs no GPU: the largest BASELINE grid needs 236 GB as a whole (three fine
    fields of 2049^2 x 2050 doubles each, the coarse levels about a seventh of that) and an eighth
    of it plus halo planes and the gathered levels per GPU on eight."""
    dh = (1e-3,) * 3
    one = mg.device_bytes((2049,) * 3, dh, 7)
    fine = 3 * 2049 * 2049 * 2050 * 8
    assert fine < one < fine * 1.16
    eight = mg.device_bytes((2049,) * 3, dh, 7, devices=list(range(8)))
    assert one / 8 < eight < one / 8 * 1.05
    assert mg.device_bytes((41,) * 3, dh, 2, flags=mg.MG_FLAG_RESERVE_EF) - mg.device_bytes((41,) * 3, dh, 2) == 41 ** 3 * 24
    assert mg.device_bytes((41,) * 3, dh, 2, flags=mg.MG_FLAG_KEEP_FINE_RESIDUAL) > mg.device_bytes((41,) * 3, dh, 2)
    with pytest.raises(mg.MgError):  # 21 planes cannot be cut into 16 slabs of whole units
        mg.device_bytes((21,) * 3, dh, 1, devices=list(range(16)))


def test_no_silent_cpu_fallback(mg):
    """Without a GPU every compute entry point must fail loudly."""
    if mg.device_count() > 0:
        pytest.skip("a GPU is present")
    import numpy as np
    with pytest.raises(mg.MgError) as e:
        mg.Handle((41, 41, 41), (1e-2,) * 3, 2)
    assert e.value.code == -4  # MG_ERR_NO_DEVICE
    with pytest.raises(mg.MgError):
        mg.op_sumsq(np.ones(8))
    with pytest.raises(mg.MgError):
        mg.op_residual(np.zeros((5, 5, 5)), np.zeros((5, 5, 5)), (1e-2,) * 3)
    with pytest.raises(mg.MgError):
        mg.op_compute_ef(np.zeros((5, 5, 5)), (1e-2,) * 3)


def _arr(mg, ptr, planes, nj, nk, i0):
    return mg.MgDevArray(ptr, planes, nj, nk, i0)


def test_level_coefficients_match_the_oracle(mg, orc):
    """1/(4^l dh^2) exactly as the reference writes it (cpp/PotentialSolver.cpp:3728-3750)."""
    dh = (0.4 / 512, 0.37 / 512, 0.41 / 512)
    for level in range(7):
        out = (C.c_double * 4)()
        assert mg.load().mg_level_coef((C.c_double * 3)(*dh), level, out) == 0
        assert tuple(out) == orc.coeffs(dh, level)
    assert mg.load().mg_dev_pitch(513) == 514 and mg.load().mg_dev_pitch(512) == 512


def test_slab_entry_points_validate_before_launching(mg):
    """Argument checks of the device-level operators run on the host, before any kernel: bad plane
    ranges, missing halo planes, misaligned pointers and mismatched twins are refused (no GPU is
    touched, so this runs anywhere)."""
    lib = mg.load()
    n = (C.c_int64 * 3)(64, 40, 40)
    c = (C.c_double * 4)(1.0, 1.0, 1.0, 6.0)
    flag = C.c_int(0)
    fake = 0x10000  # never dereferenced: every call below must fail validation first
    u = _arr(mg, fake, 20, 40, 40, 10)          # planes [10, 30)
    alt = _arr(mg, fake + 0x100000, 20, 40, 40, 10)
    rhs = _arr(mg, fake + 0x200000, 18, 40, 40, 11)
    cases = [
        # sweep planes [12, 29): needs u planes up to 31, only [10,30) stored
        lambda: lib.mg_dev_smooth_rb(None, C.byref(u), C.byref(alt), C.byref(rhs), n, c, 1.4, 1, 12, 29, 0, 0, C.byref(flag)),
        # range outside the stored planes
        lambda: lib.mg_dev_smooth_rb(None, C.byref(u), C.byref(alt), C.byref(rhs), n, c, 1.4, 1, 5, 20, 0, 0, C.byref(flag)),
        # twin with another shape
        lambda: lib.mg_dev_smooth_rb(None, C.byref(u), C.byref(_arr(mg, fake, 19, 40, 40, 10)), C.byref(rhs), n, c,
                                     1.4, 1, 12, 28, 0, 0, C.byref(flag)),
        # misaligned base pointer
        lambda: lib.mg_dev_smooth_rb(None, C.byref(_arr(mg, fake + 8, 20, 40, 40, 10)), C.byref(alt), C.byref(rhs), n, c,
                                     1.4, 1, 12, 28, 0, 0, C.byref(flag)),
        # second range shorter than the first
        lambda: lib.mg_dev_smooth_rb(None, C.byref(u), C.byref(alt), C.byref(rhs), n, c, 1.4, 1, 12, 14, 26, 27, C.byref(flag)),
        # second range with more than one sweep
        lambda: lib.mg_dev_smooth_rb(None, C.byref(u), C.byref(alt), C.byref(rhs), n, c, 1.4, 2, 12, 14, 26, 28, C.byref(flag)),
        # fused residual must start on an even plane
        lambda: lib.mg_dev_residual_restrict(None, C.byref(u), C.byref(rhs), C.byref(_arr(mg, fake, 10, 20, 20, 5)), n, c,
                                             C.c_void_p(fake), 1 << 16, 13, 20, C.byref(C.c_int64(0))),
        # prolongation without the coarse planes it interpolates from
        lambda: lib.mg_dev_prolong(None, C.byref(_arr(mg, fake, 3, 21, 21, 9)), C.byref(u), n, 12, 28, 0),
        # field kernel without the halo plane
        lambda: lib.mg_dev_compute_ef(None, C.byref(u), C.c_void_p(fake), 10, n, (C.c_double * 3)(1e-3, 1e-3, 1e-3), 10, 20),
    ]
    for k, call in enumerate(cases):
        assert call() == -1, f"case {k} was not refused"  # MG_ERR_INVALID
        assert lib.mg_last_error()
    assert lib.mg_dev_sweep_out_of_place(C.byref(_arr(mg, None, 40, 300, 300, 0))) == 1
    assert lib.mg_dev_sweep_out_of_place(C.byref(_arr(mg, None, 40, 100, 100, 0))) == 1  # by tiles instead of a march
    assert lib.mg_dev_onload_folds(C.byref(_arr(mg, None, 40, 200, 200, 0)), 1) == 1   # correction: from 128 rows
    assert lib.mg_dev_onload_folds(C.byref(_arr(mg, None, 40, 200, 200, 0)), 0) == 0   # prolongation: from 256
    assert lib.mg_dev_onload_folds(C.byref(_arr(mg, None, 40, 300, 300, 0)), 0) == 1
    # two sweeps in one launch: planes four beyond the range must be stored (eps arrays: one node more than the level)
    i3 = lambda a, b, c_: (C.c_int64 * 3)(a, b, c_)
    pairs = lambda u, rhs, n_, lo, hi: lib.mg_dev_sweeps_in_pairs(C.byref(u), C.byref(rhs), n_, lo, hi)
    assert pairs(_arr(mg, None, 301, 301, 301, 0), _arr(mg, None, 300, 300, 300, 0), i3(300, 300, 300), 0, 301) == 1  # whole array
    assert pairs(_arr(mg, None, 48, 301, 301, 100), _arr(mg, None, 46, 300, 300, 101), i3(300, 300, 300), 104, 144) == 1  # slab, four halo planes
    assert pairs(_arr(mg, None, 44, 301, 301, 102), _arr(mg, None, 42, 300, 300, 103), i3(300, 300, 300), 104, 144) == 0  # slab, two halo planes
    assert pairs(_arr(mg, None, 65, 65, 65, 0), _arr(mg, None, 64, 64, 64, 0), i3(64, 64, 64), 0, 65) == 1     # small level: by tiles
    assert pairs(_arr(mg, None, 129, 129, 129, 0), _arr(mg, None, 128, 128, 128, 0), i3(128, 128, 128), 40, 80) == 0  # 40 planes of 129^2: tiles, one sweep each
