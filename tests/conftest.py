import os
import sys

import pytest

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
