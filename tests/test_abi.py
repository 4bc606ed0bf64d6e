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
