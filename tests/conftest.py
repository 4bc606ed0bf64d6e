import os
import sys

import pytest

# PyTorch-ROCm ships its own copy of the HIP runtime.  A process that uses both torch and
# libmg_hip.so on the GPU must load torch first, so that the library binds to the runtime that is
# already there; the other way round the process holds two runtimes and the second one to
# initialise finds no device.  Some GPU tests use torch tensors as device buffers.
import torch  # noqa: F401  (before anything loads libmg_hip.so)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import mg_oracle
    mg_oracle.build()
    return mg_oracle


@pytest.fixture(scope="session")
def mg():
    """The HIP library binding.  Loading must succeed wherever tests run (the .so
    is built in-tree); GPU tests are skipped only when the machine has no GPU at all."""
    from multigrid_amd import libmg
    libmg.load()
    return libmg


@pytest.fixture(scope="session")
def gpu(mg):
    if mg.device_count() == 0:
        pytest.skip("no HIP device on this machine")
    return mg
